"""GPU end-to-end test of the experiment layer: rewire -> split -> GCN -> training_loop
-> pickles -> evaluation, the flow of the reference's experiment/save_models.py and experiment/test_performance.py."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def oracle():
    from oracle import c_oracle
    return c_oracle


def _tiny_dataset(folder, n=400, n_feat=24, n_cls=4):
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(n, 3, seed=21)
    rng = np.random.Generator(np.random.PCG64(5))
    y = rng.integers(0, n_cls, n)
    x = rng.standard_normal((n, n_feat)).astype(np.float32)
    x[np.arange(n), y] += 3.0                        # learnable: the label shows in the features
    np.savez(os.path.join(folder, 'Tiny.npz'), x=x, y=y, edge_index=ei)
    return ei, n


def test_save_models_then_test_performance(tmp_path, oracle):
    from experiment.save_models import save_models, _dump
    from experiment.test_performance import test_performance as evaluate_saved, summary_cell
    from utils.hyperparams import hyperparams
    from utils.seeds import val_seeds
    tmp = str(tmp_path)
    ei, n = _tiny_dataset(tmp)
    hp = hyperparams['Cora']                         # names outside the table train with Cora's values
    np.random.seed(3)
    torch.manual_seed(3)
    sds = save_models('Tiny', 'bfc', patience=5, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:2], epochs=40,
                      verbose=False)
    assert len(sds) == 2
    assert sorted(sds[0]) == ['layers.0.bias', 'layers.0.lin.weight', 'layers.1.bias', 'layers.1.lin.weight']
    assert sds[0]['layers.0.lin.weight'].shape == (hp['hidden_dim'], 24) and not sds[0]['layers.0.bias'].is_cuda
    with open(os.path.join(tmp, 'edge_indices', 'Tiny', 'edge_index_bfc.pk'), 'rb') as f:
        rewired = pickle.load(f)
    assert rewired.dtype == torch.int64 and not rewired.is_cuda and rewired.shape[0] == 2
    # the pickled edge list is exactly what the CPU oracle's SDRF gives for the same numpy stream
    np.random.seed(3)
    want = oracle.sdrf(ei, n, 'bfc', hp['max_iterations'], True, hp['removal_bound'], hp['tau'], nthreads=8)
    assert np.array_equal(rewired.numpy(), want)
    _dump(sds, os.path.join(tmp, 'state_dicts', 'Tiny', 'state_dicts_bfc.pk'))
    accs, mean, std = evaluate_saved('Tiny', 'bfc', test=False, data_dir=tmp, out_dir=tmp)
    assert len(accs) == 2 and all(0.5 < a <= 1.0 for a in accs), accs      # 4 classes, separable features
    test_accs, _, _ = evaluate_saved('Tiny', 'bfc', test=True, data_dir=tmp, out_dir=tmp)
    assert len(test_accs) == 2 and all(0.0 <= a <= 1.0 for a in test_accs)
    assert '+-' in summary_cell(mean, std)
    # curvature None: no rewiring, the stored edge list is the input's
    sd0 = save_models('Tiny', None, patience=3, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:1], epochs=5, verbose=False)
    with open(os.path.join(tmp, 'edge_indices', 'Tiny', 'edge_index_None.pk'), 'rb') as f:
        assert np.array_equal(pickle.load(f).numpy(), ei)
    assert len(sd0) == 1


def test_redo_rewiring_writes_one_edge_list_per_seed(tmp_path):
    from experiment.save_models import save_models
    from utils.seeds import val_seeds
    tmp = str(tmp_path)
    _tiny_dataset(tmp, n=200)
    np.random.seed(0)
    save_models('Tiny', 'augmented', patience=2, redo_rewiring=True, data_dir=tmp, out_dir=tmp, seeds=val_seeds[:2],
                epochs=3, verbose=False)
    folder = os.path.join(tmp, 'edge_indices', 'Tiny_redo_rewiring', 'augmented')
    assert sorted(os.listdir(folder)) == ['edge_index_augmented_00.pk', 'edge_index_augmented_01.pk']


@pytest.mark.gpu
def test_graphed_epochs_equal_eager_epochs():
    """experiment/training_loop.py replays the training step and the validation forward as captured HIP graphs after
    three eager epochs.  Same kernels in the same order: with dropout off (no random stream involved) the weights and
    the validation accuracies after 12 epochs are identical to 12 eager epochs; with dropout on the masks must differ
    from replay to replay (the Philox call counter lives in device memory)."""
    import copy
    import torch
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from experiment.training_loop import GraphedEpoch, evaluate, make_epoch, train
    from models.gcn import GCN
    dev = torch.device('cuda')
    ei_np, n = synthetic.powerlaw_graph(1500, 3, seed=5)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, 96, device=dev, generator=g)
    y = torch.randint(0, 5, (n,), device=dev, generator=g)
    r = torch.rand(n, device=dev, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei_np).to(dev), y=y, num_nodes=n, train_mask=r < 0.3,
                val_mask=(r >= 0.3) & (r < 0.6))

    def build(dropout):
        torch.manual_seed(3)
        model = GCN(Dataset(data, 5), hidden=[32], dropout=dropout).to(dev)
        opt = torch.optim.Adam([{'params': model.non_reg_params, 'weight_decay': 0},
                                {'params': model.reg_params, 'weight_decay': 5e-3}], lr=0.02, capturable=True)
        return model, opt

    m1, o1 = build(0.0)
    accs_eager = []
    for _ in range(12):
        train(m1, o1, data)
        accs_eager.append(evaluate(m1, data, test=False)['val_acc'])
    m2, o2 = build(0.0)
    epoch = make_epoch(m2, o2, data)
    assert isinstance(epoch, GraphedEpoch)
    accs_graph = [epoch() for _ in range(12)]
    assert epoch.train_graph is not None
    assert accs_graph == accs_eager
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    # dropout on: every replay must draw a fresh mask -> the loss sequence is not periodic / constant
    m3, o3 = build(0.5)
    epoch3 = make_epoch(m3, o3, data)
    before = None
    changed = 0
    for i in range(10):
        epoch3()
        if i >= GraphedEpoch.WARMUP + 1:
            cur = float(epoch3.loss.detach())
            changed += before is not None and cur != before
            before = cur
    assert changed >= 4
    # without a capturable optimiser the loop stays eager
    m4 = GCN(Dataset(data, 5), hidden=[32], dropout=0.0).to(dev)
    o4 = torch.optim.Adam(m4.parameters(), lr=0.01)
    assert not isinstance(make_epoch(m4, o4, data), GraphedEpoch)


@pytest.mark.parametrize('graphed', [False, True])
def test_training_loop_reproduces_the_reference_run_on_the_gpu(graphed, monkeypatch):
    """The same fixture (reference experiment/training_loop.py:22-37) through the eager epoch and through the two
    captured HIP graphs per epoch (GraphedEpoch); float64 model, so only the BLAS summation order differs from the CPU
    run that recorded it."""
    import training_loop_fixture as fx
    import experiment.training_loop as tl
    monkeypatch.setenv('DCR_HIP_GRAPH', '1' if graphed else '0')
    for case in fx.cases():
        model, data, losses, accs = fx.run_recorded(case, device='cuda:0', capturable=graphed)
        # (torch's capturable Adam keeps its step counter and bias corrections in float32 tensors — measured 4e-6 absolute
        #  in the weights after 12 steps, graphed or not: the run follows the recorded float64 trajectory to float32
        #  accuracy in the weights and exactly in every accuracy)
        fx.check(case, model, data, losses, accs, tol=1e-7, wtol=1e-4 if graphed else 1e-7, watol=2e-5 if graphed else 1e-9)
        assert (not losses) == graphed  # the graphed epoch does not go through train() at all
        assert tl.GraphedEpoch.supported(model, torch.optim.Adam(model.parameters(), capturable=True), data) == graphed


def _gcn_case(dropout, seed=3):
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models.gcn import GCN
    dev = torch.device('cuda')
    ei_np, n = synthetic.powerlaw_graph(1500, 3, seed=5)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, 96, device=dev, generator=g)
    y = torch.randint(0, 5, (n,), device=dev, generator=g)
    r = torch.rand(n, device=dev, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei_np).to(dev), y=y, num_nodes=n, train_mask=r < 0.3,
                val_mask=(r >= 0.3) & (r < 0.6))
    torch.manual_seed(seed)
    model = GCN(Dataset(data, 5), hidden=[32], dropout=dropout).to(dev)
    opt = torch.optim.Adam([{'params': model.non_reg_params, 'weight_decay': 0},
                            {'params': model.reg_params, 'weight_decay': 5e-3}], lr=0.02, capturable=True)
    return model, opt, data


def test_pair_aggregation_and_forward_pair_are_bit_identical_to_separate_passes():
    """dcr_spmm_csr_f32_pair_dev: every block as a call of its own; GCN.forward_pair: what train-mode and eval-mode forwards
    return (dropout 0, so that the two are comparable value for value)."""
    from models.gcn import gcn_norm_csr, spmm, spmm_pair
    model, _, data = _gcn_case(0.0)
    csr = gcn_norm_csr(data.edge_index, None, data.num_nodes)
    for f in (5, 16, 32, 6):
        g = torch.Generator(device='cuda').manual_seed(f)
        b2 = torch.randn(data.num_nodes, 2 * f, device='cuda', generator=g)
        bias = torch.randn(f, device='cuda', generator=g)
        both = spmm_pair(csr.rowptr, csr.col, csr.val, b2, csr.n_rows, f, bias=bias)
        assert torch.equal(both[:, :f], spmm(csr.rowptr, csr.col, csr.val, b2[:, :f].contiguous(), csr.n_rows, bias=bias))
        assert torch.equal(both[:, f:], spmm(csr.rowptr, csr.col, csr.val, b2[:, f:].contiguous(), csr.n_rows, bias=bias))
    model.train()
    lp_train, lp_eval = model.forward_pair(data)
    assert torch.equal(lp_train, model(data))
    model.eval()
    with torch.no_grad():
        assert torch.equal(lp_eval, model(data))
    model.train()
    lp_train, _ = model.forward_pair(data)
    lp_train.sum().backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    model(data).sum().backward()
    for a, p in zip(got, model.parameters()):
        assert torch.equal(a, p.grad)


@pytest.mark.parametrize('epochs,patience', [(40, 4), (9, 100), (1, 5)])
def test_one_graph_per_epoch_training_loop_equals_the_plain_loop(epochs, patience, monkeypatch):
    """training_loop through LaggedGraphedEpoch (accuracies consumed one step late, a step past the stopping epoch
    discarded) returns the weights and visits the accuracies of the plain train / evaluate loop
    (experiment/training_loop.py:22-37)."""
    import copy
    import experiment.training_loop as tl
    m1, o1, data = _gcn_case(0.0)
    best, best_w, streak, accs = 0, None, 0, []
    for _ in range(epochs):
        tl.train(m1, o1, data)
        acc = tl.evaluate(m1, data, test=False)['val_acc']
        accs.append(acc)
        if acc >= best:
            best, streak, best_w = acc, 0, copy.deepcopy(m1.state_dict())
        else:
            streak += 1
        if streak >= patience:
            break
    m2, o2, _ = _gcn_case(0.0)
    assert tl.LaggedGraphedEpoch.supported(m2, o2, data)
    seen = []
    orig = tl.LaggedGraphedEpoch.step
    monkeypatch.setattr(tl.LaggedGraphedEpoch, 'step', lambda self: seen.append(1) or orig(self))
    out = tl.training_loop(m2, o2, data, epochs, patience)
    assert out is m2
    assert len(seen) in (len(accs), len(accs) + 1)     # at most the one discarded step more
    for (k1, v1), (k2, v2) in zip(best_w.items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert tl.evaluate(m2, data, test=False)['val_acc'] == best


def test_row_selected_forward_is_the_indexed_full_forward():
    """``model(data, rows=mask)`` against ``model(data)[mask]`` (what experiment/training_loop.py:50-51,64-71 read): the same
    bits forward — every selected row is accumulated as the full aggregation accumulates it — in both modes, for masks and
    index tensors, and through ``forward_pair``; the gradients agree to rounding (the backward pass leaves out terms that
    are exactly zero, which changes the summation grouping of rows above 96 non-zeros only)."""
    from models.gcn import RowSelection, gcn_norm_csr, spmm, spmm_rows
    model, _, data = _gcn_case(0.0)
    n = data.num_nodes
    csr = gcn_norm_csr(data.edge_index, None, n)
    assert int((csr.rowptr[1:] - csr.rowptr[:-1]).max()) > 96          # both row paths of the kernel are exercised
    g = torch.Generator(device='cuda').manual_seed(7)
    sel = RowSelection(csr, data.train_mask)
    hubs_first = torch.argsort(csr.rowptr[1:] - csr.rowptr[:-1], descending=True)[:200].contiguous()
    for f in (5, 16, 32, 6):
        wide = torch.randn(n, 2 * f, device='cuda', generator=g)
        bias = torch.randn(f, device='cuda', generator=g)
        for B in (wide[:, :f], wide[:, f:], wide[:, f:].contiguous()):
            full = spmm(csr.rowptr, csr.col, csr.val, B.contiguous(), csr.n_rows, bias=bias)
            assert torch.equal(spmm_rows(csr, sel, B, bias), full[data.train_mask])
            assert torch.equal(spmm_rows(csr, RowSelection(csr, hubs_first), B, bias), full.index_select(0, hubs_first))
    assert spmm_rows(csr, RowSelection(csr, torch.zeros(n, dtype=torch.bool, device='cuda')), wide[:, :f]).shape == (0, f)
    # the transposed, column-restricted matrix against the full transposed product of a gradient that is zero elsewhere
    grad = torch.zeros(n, 16, device='cuda')
    compact = torch.randn(sel.n, 16, device='cuda', generator=g)
    grad[data.train_mask] = compact
    want = spmm(csr.rowptr_t, csr.col_t, csr.val_t, grad, csr.n_cols)
    rp, ci, va = sel.transposed()
    got = spmm(rp, ci, va, compact, csr.n_cols)
    short = (csr.rowptr_t[1:] - csr.rowptr_t[:-1]) <= 96
    assert torch.equal(got[short], want[short])                         # rows accumulated in order: zero terms change nothing
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    val_idx = data.val_mask.nonzero().squeeze(1)
    for mode in ('train', 'eval'):
        model.train(mode == 'train')
        with torch.no_grad():
            full = model(data)
            assert torch.equal(model(data, rows=data.train_mask), full[data.train_mask])
            assert torch.equal(model(data, rows=val_idx), full.index_select(0, val_idx))
    model.train()
    lp_tr, lp_ev = model.forward_pair(data, rows_train=data.train_mask, rows_eval=val_idx)
    f_tr, f_ev = model.forward_pair(data)
    assert torch.equal(lp_tr, f_tr[data.train_mask]) and torch.equal(lp_ev, f_ev.index_select(0, val_idx))
    y = data.y[data.train_mask]
    model.zero_grad()
    torch.nn.functional.nll_loss(lp_tr, y).backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    torch.nn.functional.nll_loss(f_tr[data.train_mask], y).backward()
    for a, p in zip(got, model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-4, atol=1e-7), (a - p.grad).abs().max()


def test_row_selection_with_repeated_and_negative_indices():
    """An index tensor that names a node twice (or from the end): ``model(data, rows=idx)`` still equals
    ``model(data)[idx]`` row for row AND its backward adds the repeats' gradients as indexing does (round-3 advisor: the
    column map of the restricted transpose keeps one position per node and silently dropped the others); out-of-range
    indices raise like indexing does."""
    model, _, data = _gcn_case(0.0)
    n = data.num_nodes
    base = data.train_mask.nonzero().squeeze(1)[:40]
    idx = torch.cat([base, base[:7], base[3:5], torch.tensor([-1, -n, n - 1], device=base.device)]).contiguous()
    for mode in ('train', 'eval'):
        model.train(mode == 'train')
        with torch.no_grad():
            assert torch.equal(model(data, rows=idx), model(data)[idx])
    model.train()
    w = torch.randn(idx.numel(), data.y.max().item() + 1, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
    model.zero_grad()
    (model(data, rows=idx) * w).sum().backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    (model(data)[idx] * w).sum().backward()
    for a, p in zip(got, model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-4, atol=1e-6), (a - p.grad).abs().max()
    lp_tr, lp_ev = model.forward_pair(data, rows_train=idx, rows_eval=idx)
    f_tr, f_ev = model.forward_pair(data)
    assert torch.equal(lp_tr, f_tr[idx]) and torch.equal(lp_ev, f_ev[idx])
    for bad in (torch.tensor([0, n], device='cuda'), torch.tensor([-n - 1], device='cuda')):
        with pytest.raises(IndexError):
            model(data, rows=bad)


def test_epochs_on_selected_rows_follow_the_full_output_epochs(monkeypatch):
    """Twelve epochs with the last aggregation evaluated at the split rows against twelve epochs on the full output
    (DCR_GCN_ALL_ROWS=1): the same accuracies, weights equal to rounding."""
    import experiment.training_loop as tl
    runs = {}
    for all_rows in ('1', '0'):
        monkeypatch.setenv('DCR_GCN_ALL_ROWS', all_rows)
        model, opt, data = _gcn_case(0.0)
        accs = []
        for _ in range(12):
            tl.train(model, opt, data)
            accs.append(tl.evaluate(model, data, test=False)['val_acc'])
        data.test_mask = ~(data.train_mask | data.val_mask)
        both = tl.evaluate(model, data, test=True)          # (two splits: the full output, indexed twice)
        assert both['val_acc'] == accs[-1] and 0 <= both['test_acc'] <= 1
        runs[all_rows] = (accs, [p.detach().clone() for p in model.parameters()])
    assert runs['0'][0] == runs['1'][0]
    for a, b in zip(runs['0'][1], runs['1'][1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def test_loss_and_accuracy_kernels_against_the_stock_ops():
    """dcr_nll_picked_mean_* and dcr_count_argmax_equal_f32_dev (the masked NLL and the arg-max accuracy of
    experiment/training_loop.py:51,64-71 on the selected rows): the gradient bit for bit what F.nll_loss gives, the loss to
    rounding, the count exactly — ties resolved to the first maximum and a NaN counted as the maximum, as torch.max does."""
    import torch.nn.functional as F
    from experiment.training_loop import _PickedMean, _count_correct
    g = torch.Generator(device='cuda').manual_seed(4)
    for m, c in ((1, 3), (1000, 16), (99703, 16), (4097, 7)):
        logits = torch.randn(m, c, device='cuda', generator=g)
        y = torch.randint(0, c, (m,), device='cuda', generator=g)
        a = logits.clone().requires_grad_(True)
        b = logits.clone().requires_grad_(True)
        la = _PickedMean.apply(F.log_softmax(a, dim=1), y)
        lb = F.nll_loss(F.log_softmax(b, dim=1), y)
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb)))
        (la * 1.0).backward()
        lb.backward()
        assert torch.equal(a.grad, b.grad), (m, c)
        lp = F.log_softmax(logits, dim=1)
        assert int(_count_correct(lp, y)) == int(lp.max(1)[1].eq(y).sum())
    # ties and NaNs
    lp = torch.tensor([[0.5, 0.5, 0.1], [0.1, 0.7, 0.7], [float('nan'), 1.0, 2.0], [1.0, float('nan'), float('nan')], [0.0, 0.0, 0.0]],
                      device='cuda')
    for y in ([0, 1, 0, 1, 0], [1, 2, 2, 2, 2], [0, 1, 2, 0, 1]):
        yt = torch.tensor(y, device='cuda')
        assert int(_count_correct(lp, yt)) == int(lp.max(1)[1].eq(yt).sum()), y


def test_head_kernels_against_the_stock_ops():
    """dcr_head_fwd_f32_dev / dcr_head_bwd_f32_dev (round 5: models/gcn.py:44 log_softmax + experiment/training_loop.py:51 nll_loss
    on the training rows and :64-71 the arg-max accuracy on the evaluated rows, one kernel per direction from the RAW outputs of
    the last aggregation): loss and gradient within float32 rounding of the stock ops, the bias gradient = the column sums of
    that gradient, the count exactly (ties to the first maximum, a NaN counts as the maximum), every call reproducible bit for
    bit; and GCN.forward_head against forward_pair + the stock ops on a model."""
    import torch.nn.functional as F
    from models import gcn
    from models.gcn import RowSelection, _AggregateRowsHead, gcn_norm_csr
    model, _, data = _gcn_case(0.0)
    n = data.num_nodes
    csr = gcn_norm_csr(data.edge_index, None, n)
    g = torch.Generator(device='cuda').manual_seed(4)
    s_tr, s_ev = RowSelection(csr, data.train_mask), RowSelection(csr, data.val_mask)
    y_tr, y_ev = data.y[data.train_mask].contiguous(), data.y[data.val_mask].contiguous()
    for c in (5, 7, 16, 32):
        both = torch.randn(n, 2 * c, device='cuda', generator=g)
        bias = torch.randn(c, device='cuda', generator=g)
        ytr, yev = y_tr % c, y_ev % c
        z_tr = both[:, :c].clone().requires_grad_(True) if c % 4 else None
        a = both.clone().requires_grad_(True)
        ba = bias.clone().requires_grad_(True)
        za, ze = (a[:, :c], a[:, c:]) if c % 4 == 0 else (z_tr, both[:, c:].contiguous())
        loss, correct = _AggregateRowsHead.apply(za, ze, ba, csr, s_tr, s_ev, ytr, yev)
        loss2, correct2 = _AggregateRowsHead.apply(za.detach(), ze, ba.detach(), csr, s_tr, s_ev, ytr, yev)
        assert torch.equal(loss.detach(), loss2) and torch.equal(correct, correct2)
        b = both.clone().requires_grad_(True)
        bb = bias.clone().requires_grad_(True)
        o_tr = gcn.aggregate_rows(b[:, :c], bb, csr, s_tr)
        o_ev = gcn.spmm_rows(csr, s_ev, both[:, c:], bias)
        want = F.nll_loss(F.log_softmax(o_tr, dim=1), ytr)
        assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
        assert int(correct) == int(o_ev.max(1)[1].eq(yev).sum())
        loss.backward()
        want.backward()
        got_z = (a.grad[:, :c] if c % 4 == 0 else z_tr.grad)
        scale = b.grad[:, :c].abs().max().item()
        assert (got_z - b.grad[:, :c]).abs().max().item() <= 2e-6 * scale
        assert (ba.grad - bb.grad).abs().max().item() <= 1e-5 * bb.grad.abs().max().item()
        only_tr = _AggregateRowsHead.apply(za.detach(), None, bias, csr, s_tr, None, ytr, None)
        only_ev = _AggregateRowsHead.apply(None, ze, bias, csr, None, s_ev, None, yev)
        assert torch.equal(only_tr[0], loss.detach()) and only_tr[1] is None and torch.equal(only_ev[1], correct) and only_ev[0] is None
    # ties and NaNs in the evaluated rows
    from dcr import _lib
    import ctypes
    lp = torch.tensor([[0.5, 0.5, 0.1], [0.1, 0.7, 0.7], [float('nan'), 1.0, 2.0], [1.0, float('nan'), float('nan')], [0.0, 0.0, 0.0]],
                      device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    ws = gcn._head_workspace(lp.device, st)
    for y in ([0, 1, 0, 1, 0], [1, 2, 2, 2, 2], [0, 1, 2, 0, 1]):
        yt = torch.tensor(y, device='cuda')
        out = torch.empty((), dtype=torch.int64, device='cuda')
        _lib.check(_lib.lib().dcr_head_fwd_f32_dev(None, 3, None, 0, lp.data_ptr(), 3, yt.data_ptr(), 5, 3, None, out.data_ptr(), ws.data_ptr(),
                                                   ws.numel() * 8, ctypes.c_void_p(st)))
        assert int(out) == int(lp.max(1)[1].eq(yt).sum()), y
    # the model: forward_head = forward_pair + the stock ops
    model.train()
    tr_idx, ev_idx = data.train_mask.nonzero().squeeze(1), data.val_mask.nonzero().squeeze(1)
    loss, correct = model.forward_head(data, rows_train=tr_idx, y_train=y_tr, rows_eval=ev_idx, y_eval=y_ev)
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    lp_tr, lp_ev = model.forward_pair(data, rows_train=tr_idx, rows_eval=ev_idx)
    want = F.nll_loss(lp_tr, y_tr)
    want.backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want))) and int(correct) == int(lp_ev.max(1)[1].eq(y_ev).sum())
    for a, p in zip(got, model.parameters()):
        assert (a - p.grad).abs().max().item() <= 1e-5 * max(p.grad.abs().max().item(), 1e-12)
    model.zero_grad()
    l_only = model.forward_head(data, rows_train=tr_idx, y_train=y_tr)[0]
    model.eval()
    with torch.no_grad():
        c_only = model.forward_head(data, rows_eval=ev_idx, y_eval=y_ev)[1]
    assert torch.equal(l_only.detach(), loss.detach()) and torch.equal(c_only, correct)


def test_forward_head_on_a_deeper_model():
    """GCN.forward_head with three hidden layers (Pubmed's depth, utils/hyperparams.py:23-30): loss, count and gradients within
    float32 rounding of forward_pair + the stock ops; a captured epoch of it runs."""
    import torch.nn.functional as F
    from dcr.data import Dataset
    from experiment.training_loop import make_epoch, LaggedGraphedEpoch
    from models.gcn import GCN
    _, _, data = _gcn_case(0.0)
    torch.manual_seed(4)
    model = GCN(Dataset(data, 5), hidden=[64, 64, 64], dropout=0.0).to('cuda')
    tr_idx, ev_idx = data.train_mask.nonzero().squeeze(1), data.val_mask.nonzero().squeeze(1)
    y_tr, y_ev = data.y[tr_idx].contiguous(), data.y[ev_idx].contiguous()
    model.train()
    loss, correct = model.forward_head(data, rows_train=tr_idx, y_train=y_tr, rows_eval=ev_idx, y_eval=y_ev)
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    lp_tr, lp_ev = model.forward_pair(data, rows_train=tr_idx, rows_eval=ev_idx)
    want = F.nll_loss(lp_tr, y_tr)
    want.backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want))) and int(correct) == int(lp_ev.max(1)[1].eq(y_ev).sum())
    for a, p in zip(got, model.parameters()):
        assert (a - p.grad).abs().max().item() <= 1e-5 * max(p.grad.abs().max().item(), 1e-12)
    # (a fresh model for the captured epoch: this one has run eager backward passes on the default stream)
    torch.manual_seed(4)
    model2 = GCN(Dataset(data, 5), hidden=[64, 64, 64], dropout=0.5).to('cuda')
    opt = torch.optim.Adam(model2.parameters(), lr=0.01, capturable=True)
    epoch = make_epoch(model2, opt, data, lagged=True)
    assert isinstance(epoch, LaggedGraphedEpoch)
    accs = [epoch() for _ in range(8)]
    assert all(0.0 <= a <= 1.0 for a in accs)


def test_dropout_decisions_drawn_ahead_give_the_same_epochs(monkeypatch):
    """``DCR_DROPOUT_AHEAD=1`` (models/gcn.py::_DropoutAhead: the next epoch's dropout decisions drawn on a side stream of the
    captured epoch by dcr_dropout_words_dev, F.dropout of models/gcn.py:40): the same parameters bit for bit as drawing in line,
    after eager and replayed epochs; the decisions in the buffer are the ones for the counter's next value."""
    from dcr.data import Dataset
    from experiment.training_loop import make_epoch, LaggedGraphedEpoch
    from models import gcn
    _, _, data = _gcn_case(0.5)
    dev = data.x.device

    def run(ahead):
        monkeypatch.setenv('DCR_DROPOUT_AHEAD', '1' if ahead else '0')
        gcn._AHEAD.clear()
        gcn._dropout_counter(dev).zero_()
        torch.manual_seed(8)
        model = gcn.GCN(Dataset(data, 5), hidden=[64], dropout=0.5).to(dev)
        opt = torch.optim.Adam([{'params': model.non_reg_params, 'weight_decay': 0},
                                {'params': model.reg_params, 'weight_decay': 5e-3}], lr=0.02, capturable=True)
        epoch = make_epoch(model, opt, data, lagged=True)
        assert isinstance(epoch, LaggedGraphedEpoch)
        accs = [epoch() for _ in range(7)]
        torch.cuda.synchronize()
        return [p.detach().clone() for p in model.parameters()], [float(a) for a in accs]
    p_off, a_off = run(False)
    assert not gcn._AHEAD
    p_on, a_on = run(True)
    (ahead,) = gcn._AHEAD.values()
    assert int(ahead.words[-8].item()) == int(gcn._dropout_counter(dev).item()) >= 7   # drawn for the next call
    assert a_on == a_off
    for a, b in zip(p_on, p_off):
        assert torch.equal(a, b)
    gcn._AHEAD.clear()


def test_one_launch_adam_follows_torch_adam():
    """experiment/adam.py::OneLaunchAdam (dcr_adam_step_f32_dev, round 5) against torch.optim.Adam on the reference's wiring
    (save_models.py:78-82: two groups, weight decay on one): parameters and moments within float32 rounding after 25 steps,
    the step counter on the device; and inside a captured HIP graph (the counter advances with every replay)."""
    from experiment.adam import OneLaunchAdam
    g = torch.Generator(device='cuda').manual_seed(6)
    shapes = [(128, 256), (128,), (16, 128), (16,)]
    init = [torch.randn(*s, device='cuda', generator=g) * 0.1 for s in shapes]
    grads = [[torch.randn(*s, device='cuda', generator=g) for s in shapes] for _ in range(25)]

    def build(cls, **kw):
        ps = [t.clone().requires_grad_(True) for t in init]
        return ps, cls([{'params': ps[2:], 'weight_decay': 0}, {'params': ps[:2], 'weight_decay': 5e-3}], lr=0.02, **kw)
    pa, oa = build(OneLaunchAdam)
    pb, ob = build(torch.optim.Adam)
    for step in grads:
        for p, q, gr in zip(pa, pb, step):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    assert float(oa._step) == 25.0
    for p, q in zip(pa, pb):
        assert (p - q).abs().max().item() <= 2e-6 * max(1.0, q.abs().max().item())
        assert (oa.state[p]['exp_avg'] - ob.state[q]['exp_avg']).abs().max().item() <= 1e-6
        assert (oa.state[p]['exp_avg_sq'] - ob.state[q]['exp_avg_sq']).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        OneLaunchAdam([{'params': [init[0].clone().requires_grad_(True)], 'lr': 0.1}, {'params': [init[1].clone().requires_grad_(True)]}], lr=0.02)
    # captured: three replays = three more steps
    pc, oc = build(OneLaunchAdam)
    static = [torch.zeros_like(t) for t in init]
    for p, gr in zip(pc, static):
        p.grad = gr
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for gr, src in zip(static, grads[0]):
            gr.copy_(src)
        oc.step()                                   # (eager warm-up: the state comes into being)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        oc.step()
    for k in (1, 2, 3):
        for gr, src in zip(static, grads[k]):
            gr.copy_(src)
        graph.replay()
    torch.cuda.synchronize()
    assert float(oc._step) == 4.0
    pd, od = build(OneLaunchAdam)
    for k in range(4):
        for p, gr in zip(pd, grads[k]):
            p.grad = gr.clone()
        od.step()
    for p, q in zip(pc, pd):
        assert torch.equal(p, q)


def test_two_row_lists_in_one_launch():
    """dcr_spmm_csr_rows2_f32_dev (training rows of the first column block, validation rows of the second) against two
    dcr_spmm_csr_rows_f32_dev calls: the same bits, with and without a bias, including empty lists on either side."""
    from models.gcn import RowSelection, _AggregateRowsPair, gcn_norm_csr, spmm_rows
    model, _, data = _gcn_case(0.0)
    n = data.num_nodes
    csr = gcn_norm_csr(data.edge_index, None, n)
    g = torch.Generator(device='cuda').manual_seed(9)
    empty = torch.zeros(n, dtype=torch.bool, device='cuda')
    for f in (16, 8, 4):
        both = torch.randn(n, 2 * f, device='cuda', generator=g)
        bias = torch.randn(f, device='cuda', generator=g)
        for m_tr, m_ev in ((data.train_mask, data.val_mask), (data.val_mask, data.train_mask), (empty, data.val_mask), (data.train_mask, empty)):
            s_tr, s_ev = RowSelection(csr, m_tr), RowSelection(csr, m_ev)
            for b in (bias, None):
                o_tr, o_ev = _AggregateRowsPair.apply(both[:, :f], both[:, f:], b, csr, s_tr, s_ev)
                assert torch.equal(o_tr, spmm_rows(csr, s_tr, both[:, :f], b)) and torch.equal(o_ev, spmm_rows(csr, s_ev, both[:, f:], b))
