"""Full-batch training driver — call surface of the reference's experiment/training_loop.py:10-75.

``training_loop(model, optimizer, data, epochs, patience)``, ``train(model, optimizer, data)`` and
``evaluate(model, data, test)`` behave as the reference's do:

  * one epoch = one optimisation step on the training mask, then a validation forward pass;
  * the checkpoint moves forward whenever validation accuracy is >= the best so far (ties included);
  * after ``patience`` epochs without that, or after the last epoch, the best weights are restored;
  * ``evaluate`` reports ``val_acc`` (and ``test_acc`` when ``test`` is true) from arg-max predictions.

Any ``torch.nn.Module`` mapping ``data`` to log-probabilities works; the masks are read with
``data[f'{split}_mask']`` item access exactly as the reference does.

On the MI355X an epoch of a citation-sized graph is launch-bound (~60 small kernels and a handful of host syncs for a
millisecond of work), so ``training_loop`` runs the first epochs eagerly and then replays the training step and the
validation forward as two captured HIP graphs (``GraphedEpoch``): same kernels, same order, same arithmetic.  It does
so when the optimiser was built with ``capturable=True`` (``experiment/save_models.py`` does) and everything lives on
the GPU; otherwise, and always in ``train`` / ``evaluate`` themselves, execution is eager.  ``DCR_HIP_GRAPH=0`` disables it.
"""
import copy
import os

import torch
import torch.nn.functional as F


def train(model, optimizer, data):
    """One optimisation step on the training nodes; returns the loss value (training_loop.py:40-54)."""
    model.train()
    optimizer.zero_grad()
    log_probs = model(data)
    mask = data.train_mask
    loss = F.nll_loss(log_probs[mask], data.y[mask])
    loss.backward()
    optimizer.step()
    return loss.item()


def _accuracy(log_probs, labels, mask):
    predicted = log_probs[mask].max(1)[1]
    return predicted.eq(labels[mask]).sum().item() / mask.sum().item()


def evaluate(model, data, test):
    """Accuracy on the validation split, plus the test split when ``test`` is true (training_loop.py:57-75)."""
    model.eval()
    with torch.no_grad():
        log_probs = model(data)
    splits = ('val', 'test') if test else ('val',)
    return {f'{split}_acc': _accuracy(log_probs, data.y, data[f'{split}_mask']) for split in splits}


class GraphedEpoch:
    """One epoch (training step, then validation accuracy) of a fixed (model, optimizer, data) triple.  The first
    ``WARMUP`` calls run eagerly on a side stream (they are real epochs: library handles, caches and the optimiser state
    come into being there); the next call captures the two HIP graphs; every later call is two graph launches and one
    host sync for the accuracy.  Boolean-mask indexing (a host sync and a data-dependent shape) is replaced by index
    tensors computed once."""

    WARMUP = 3

    def __init__(self, model, optimizer, data):
        self.model, self.optimizer, self.data = model, optimizer, data
        self.calls = 0
        self.train_graph = self.eval_graph = None
        self.stream = torch.cuda.Stream(device=data.x.device)
        self.train_idx = data.train_mask.nonzero().squeeze(1)
        self.val_idx = data['val_mask'].nonzero().squeeze(1)
        self.y_train = data.y.index_select(0, self.train_idx)
        self.y_val = data.y.index_select(0, self.val_idx)
        self.n_val = int(self.val_idx.numel())

    @staticmethod
    def supported(model, optimizer, data):
        if os.environ.get('DCR_HIP_GRAPH', '1') == '0' or not torch.cuda.is_available():
            return False
        x = getattr(data, 'x', None)
        if x is None or not x.is_cuda or not all(p.is_cuda for p in model.parameters()):
            return False
        if not all(g.get('capturable', False) for g in optimizer.param_groups):
            return False
        return int(data['val_mask'].sum()) > 0 and int(data.train_mask.sum()) > 0

    def _train_step(self):
        log_probs = self.model(self.data)
        loss = F.nll_loss(log_probs.index_select(0, self.train_idx), self.y_train)
        loss.backward()
        self.optimizer.step()
        return loss

    def _val_correct(self):
        with torch.no_grad():
            log_probs = self.model(self.data)
        return log_probs.index_select(0, self.val_idx).max(1)[1].eq(self.y_val).sum()

    def __call__(self):
        """Runs one epoch; returns the validation accuracy."""
        self.calls += 1
        cur = torch.cuda.current_stream(self.data.x.device)
        if self.calls <= self.WARMUP:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self.model.train()
                self.optimizer.zero_grad(set_to_none=True)
                self._train_step()
                self.model.eval()
                correct = self._val_correct()
            cur.wait_stream(self.stream)
            return correct.item() / self.n_val
        if self.train_graph is None:
            self.model.train()
            self.optimizer.zero_grad(set_to_none=True)
            self.train_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.train_graph, stream=self.stream):
                self.loss = self._train_step()
            self.model.eval()
            self.eval_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.eval_graph, stream=self.stream):
                self.correct = self._val_correct()
        self.train_graph.replay()
        self.eval_graph.replay()
        return self.correct.item() / self.n_val


def make_epoch(model, optimizer, data):
    """A callable that runs one epoch (training step + validation forward) and returns the validation accuracy:
    graph replay where supported, the eager ``train`` / ``evaluate`` pair otherwise."""
    if GraphedEpoch.supported(model, optimizer, data):
        return GraphedEpoch(model, optimizer, data)

    def eager():
        train(model, optimizer, data)
        return evaluate(model, data, test=False)['val_acc']
    return eager


def training_loop(model, optimizer, data, epochs, patience):
    """Train with early stopping on validation accuracy; returns the model holding the best weights
    (training_loop.py:10-37)."""
    best_acc, best_weights, since_best = 0, None, 0
    epoch = make_epoch(model, optimizer, data)
    for _ in range(epochs):
        val_acc = epoch()
        if val_acc >= best_acc:  # ties advance the checkpoint, as in the reference
            best_acc, since_best = val_acc, 0
            best_weights = copy.deepcopy(model.state_dict())
        else:
            since_best += 1
        if since_best >= patience:
            break
    model.load_state_dict(best_weights)
    return model
