"""Shared by the CPU and GPU tests of experiment/training_loop.py: rebuilds the inputs of
tests/golden/training_loop_reference.json (recorded by running the reference's own experiment/training_loop.py:10-75,
tools/make_golden_experiment.py) and replays them through this package's driver."""
import json
import os

import torch

from dcr.data import Data

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'training_loop_reference.json')
LOSS_RTOL = 1e-9   # float64 model: what is left is summation order inside the BLAS of the machine at hand


def cases():
    with open(GOLDEN) as f:
        return json.load(f)['cases']


class OnX(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, data):
        return self.net(data.x)


def build(case, device='cpu', capturable=False):
    net = torch.nn.Sequential(torch.nn.Linear(case['n_feat'], case['hidden']), torch.nn.ReLU(),
                              torch.nn.Linear(case['hidden'], case['n_cls']), torch.nn.LogSoftmax(dim=1)).double()
    model = OnX(net)
    model.load_state_dict({k: torch.tensor(v, dtype=torch.float64) for k, v in case['init'].items()})
    model.to(device)
    n = case['n']

    def mask(ix):
        m = torch.zeros(n, dtype=torch.bool)
        m[torch.tensor(ix, dtype=torch.long)] = True
        return m
    data = Data(x=torch.tensor(case['x'], dtype=torch.float64), y=torch.tensor(case['y']), train_mask=mask(case['train']),
                val_mask=mask(case['val']), test_mask=mask(case['test']), num_nodes=n).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=case['lr'], weight_decay=case['weight_decay'], capturable=capturable)
    return model, opt, data


def run_recorded(case, device='cpu', capturable=False):
    """experiment.training_loop.training_loop with wrappers recording what each epoch returned."""
    import experiment.training_loop as tl
    model, opt, data = build(case, device, capturable)
    losses, accs = [], []
    orig = tl.train, tl.evaluate, tl.GraphedEpoch.__call__

    def rec_train(m, o, d):
        v = orig[0](m, o, d)
        losses.append(v)
        return v

    def rec_eval(m, d, test):
        v = orig[1](m, d, test)
        accs.append(v['val_acc'])
        return v

    def rec_graphed(self):
        v = orig[2](self)
        accs.append(v)
        return v
    tl.train, tl.evaluate, tl.GraphedEpoch.__call__ = rec_train, rec_eval, rec_graphed
    try:
        out = tl.training_loop(model, opt, data, case['epochs'], case['patience'])
    finally:
        tl.train, tl.evaluate, tl.GraphedEpoch.__call__ = orig
    assert out is model
    return model, data, losses, accs


def check(case, model, data, losses, accs, tol=LOSS_RTOL, wtol=1e-7, watol=1e-9):
    import experiment.training_loop as tl
    assert len(accs) == case['epochs_run']
    assert accs == case['val_accs']                      # counts over small masks: exact
    if losses:
        want = [float.fromhex(h) for h in case['losses']]
        assert len(losses) == len(want)
        for e, (a, b) in enumerate(zip(losses, want)):
            assert abs(a - b) <= tol * max(1.0, abs(b)), (e, a, b)
    for k, v in model.state_dict().items():              # the weights of the best epoch (>= rule: the LAST maximum)
        w = torch.tensor(case['final'][k], dtype=torch.float64)
        assert torch.allclose(v.cpu(), w, rtol=wtol, atol=watol), k
    ev = tl.evaluate(model, data, True)
    assert ev['val_acc'] == case['final_val_acc'] == case['val_accs'][case['best_epoch']]
    assert ev['test_acc'] == case['final_test_acc']
