"""GCN path: call surface, normalisation, HIP SpMM numerics (fp32, tolerance 1e-5 as BASELINE.json's north_star
states), autograd, the training driver and the 2-rank data-parallel path (gloo on CPU).

The CPU tests select the plain-torch aggregation EXPLICITLY (models.gcn.set_aggregate_backend('torch')); the
product default is the HIP kernel and refuses CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch

TOL = 1e-5


def _toy(n=60, f=12, c=4, seed=0):
    from dcr import synthetic
    from dcr.data import Data, Dataset
    ei, n = synthetic.powerlaw_graph(n, 3, seed=seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, generator=g)
    y = torch.randint(0, c, (n,), generator=g)
    perm = torch.randperm(n, generator=g)
    masks = {}
    for name, sl in (('train', slice(0, n // 2)), ('val', slice(n // 2, 3 * n // 4)), ('test', slice(3 * n // 4, n))):
        m = torch.zeros(n, dtype=torch.bool)
        m[perm[sl]] = True
        masks[f'{name}_mask'] = m
    data = Data(x=x, edge_index=torch.from_numpy(ei), y=y, num_nodes=n, **masks)
    return data, Dataset(data, c)


@pytest.fixture
def torch_backend():
    from models import gcn
    gcn.set_aggregate_backend('torch')
    yield
    gcn.set_aggregate_backend('hip')


def test_call_surface_and_state_dict_keys():
    from models.gcn import GCN
    data, ds = _toy()
    m = GCN(ds, hidden=[16], dropout=0.3)
    assert list(m.state_dict().keys()) == ['layers.0.bias', 'layers.0.lin.weight', 'layers.1.bias',
                                           'layers.1.lin.weight']
    assert m.state_dict()['layers.0.lin.weight'].shape == (16, 12)
    assert m.state_dict()['layers.1.lin.weight'].shape == (4, 16)
    assert {id(p) for p in m.reg_params} == {id(p) for p in m.layers[0].parameters()}
    assert {id(p) for p in m.non_reg_params} == {id(p) for p in m.layers[1].parameters()}
    assert float(m.layers[0].bias.abs().sum()) == 0.0
    a = (6.0 / (12 + 16)) ** 0.5
    assert float(m.layers[0].lin.weight.abs().max()) <= a
    m2 = GCN(ds, hidden=[16], dropout=0.3)
    m2.load_state_dict(m.state_dict())                                   # test_performance.py:72-73
    m.reset_parameters()


def test_gcn_norm_matches_dense():
    from models.gcn import gcn_norm_csr
    data, _ = _toy()
    n = data.num_nodes
    csr = gcn_norm_csr(data.edge_index, None, n)
    A = torch.zeros(n, n, dtype=torch.float64)
    A[data.edge_index[1], data.edge_index[0]] = 1
    A += torch.eye(n, dtype=torch.float64)
    dinv = A.sum(1).pow(-0.5)
    Ah = dinv[:, None] * A * dinv[None, :]
    dense = torch.zeros(n, n, dtype=torch.float64)
    rows = torch.repeat_interleave(torch.arange(n), csr.rowptr[1:] - csr.rowptr[:-1])
    dense[rows, csr.col.long()] = csr.val.double()
    assert (dense - Ah).abs().max() < 1e-6
    dense_t = torch.zeros(n, n, dtype=torch.float64)
    rows_t = torch.repeat_interleave(torch.arange(n), csr.rowptr_t[1:] - csr.rowptr_t[:-1])
    dense_t[rows_t, csr.col_t.long()] = csr.val_t.double()
    assert (dense_t - Ah.t()).abs().max() < 1e-6
    # isolated node: its self loop has weight 1
    ei = torch.tensor([[0, 1], [1, 0]])
    c3 = gcn_norm_csr(ei, None, 3)
    assert c3.val[c3.rowptr[2]:c3.rowptr[3]].tolist() == [1.0]


def test_forward_backward_vs_dense_reference_cpu(torch_backend):
    from models.gcn import GCN, dense_reference_logits
    data, ds = _toy()
    torch.manual_seed(1)
    m = GCN(ds, hidden=[16], dropout=0.5)
    with torch.no_grad():
        for layer in m.layers:
            layer.bias.uniform_(-0.1, 0.1)
    m.eval()
    got = m(data)
    want = dense_reference_logits(m, data.x, data.edge_index, data.num_nodes)
    assert (got.double() - want).abs().max() < TOL
    loss = torch.nn.functional.nll_loss(got[data.train_mask], data.y[data.train_mask])
    loss.backward()
    g_got = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    want32 = dense_reference_logits(m, data.x, data.edge_index, data.num_nodes).float()
    torch.nn.functional.nll_loss(want32[data.train_mask], data.y[data.train_mask]).backward()
    for a, p in zip(g_got, m.parameters()):
        assert (a - p.grad).abs().max() < 1e-5


def test_training_loop_semantics(torch_backend):
    from experiment.training_loop import evaluate, train, training_loop
    from models.gcn import GCN
    data, ds = _toy(n=80)
    torch.manual_seed(3)
    m = GCN(ds, hidden=[8], dropout=0.0)
    opt = torch.optim.Adam([{'params': m.non_reg_params, 'weight_decay': 0}, {'params': m.reg_params, 'weight_decay': 5e-4}],
                           lr=0.05)                                       # save_models.py:78-82
    l0 = train(m, opt, data)
    assert np.isfinite(l0)
    r = evaluate(m, data, test=True)
    assert set(r) == {'val_acc', 'test_acc'} and 0 <= r['val_acc'] <= 1
    assert set(evaluate(m, data, test=False)) == {'val_acc'}

    # early stopping: a model whose validation accuracy peaks at epoch 2 must come back with epoch-2 weights
    class Scripted(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
            self.step = 0
            self.acc = [0.2, 0.5, 0.5, 0.4, 0.3, 0.3, 0.9]

        def forward(self, d):
            n = d.y.shape[0]
            k = int(round(self.acc[min(self.step, len(self.acc) - 1)] * int(d.val_mask.sum())))
            logp = torch.full((n, 2), -5.0) + self.w * 0
            right = d.y.clone()
            idx = d.val_mask.nonzero().flatten()
            wrong = idx[k:]
            right[wrong] = 1 - right[wrong]
            logp[torch.arange(n), right] = 0.0
            return logp

    class Opt:
        def __init__(self, m): self.m = m
        def zero_grad(self): pass
        def step(self):
            self.m.step += 1
            with torch.no_grad():
                self.m.w += 1.0

    from dcr.data import Data
    y = torch.randint(0, 2, (40,))
    tm = torch.zeros(40, dtype=torch.bool); tm[:10] = True
    vm = torch.zeros(40, dtype=torch.bool); vm[10:30] = True
    d = Data(y=y, train_mask=tm, val_mask=vm, num_nodes=40)
    s = Scripted()
    out = training_loop(s, Opt(s), d, epochs=50, patience=3)
    # validation accuracies seen after each step: .5 .5 .4 .3 .3 -> best (ties move forward) is after step 2,
    # three non-improving epochs follow, so training stops after step 5 and restores w == 2
    assert out is s and float(s.w) == 2.0 and s.step == 5


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from models import gcn
    from models.gcn import GCN
    from models.gcn_dp import ShardedGCN
    gcn.set_aggregate_backend('torch')
    data, ds = _toy(n=61, seed=4)                                         # 61: uneven blocks
    torch.manual_seed(7)
    base = GCN(ds, hidden=[8], dropout=0.0)
    ref = GCN(ds, hidden=[8], dropout=0.0)
    ref.load_state_dict(base.state_dict())
    sh = ShardedGCN(base, data.edge_index, data.num_nodes)
    xl, yl, tl = sh.shard(data.x), sh.shard(data.y), sh.shard(data.train_mask)
    n_train = int(data.train_mask.sum())
    ref.eval(); sh.eval()
    full = ref(data)
    local = sh(xl)
    err_fwd = (local[:sh.owned.shape[0]] - full[sh.owned]).abs().max().item()   # this rank's nodes, in its block order
    opt = torch.optim.SGD(base.parameters(), lr=0.1)
    sh.train_step(opt, xl, yl, tl, n_train)
    ref.train()
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    ropt.zero_grad()
    torch.nn.functional.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
    ropt.step()
    err_w = max((a - b).abs().max().item() for a, b in zip(base.parameters(), ref.parameters()))
    acc = sh.eval_correct(xl, yl, sh.shard(data.val_mask))
    ref.eval()
    with torch.no_grad():
        lp = ref(data)
    acc_ref = (lp[data.val_mask].argmax(1) == data.y[data.val_mask]).float().mean().item()
    ret[rank] = (err_fwd, err_w, abs(acc - acc_ref))
    dist.destroy_process_group()


def _dp_pair_worker(rank, world, port, ret):
    """train_eval_step (one pass for the training step and the evaluation of the weights it starts from) against the
    single-process model doing evaluate-then-train."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from models import gcn
    from models.gcn import GCN
    from models.gcn_dp import ShardedGCN
    gcn.set_aggregate_backend('torch')
    data, ds = _toy(n=61, seed=4)
    torch.manual_seed(7)
    base = GCN(ds, hidden=[8], dropout=0.0)
    ref = GCN(ds, hidden=[8], dropout=0.0)
    ref.load_state_dict(base.state_dict())
    sh = ShardedGCN(base, data.edge_index, data.num_nodes)
    xl, yl = sh.shard(data.x), sh.shard(data.y)
    tl, vl = sh.shard(data.train_mask), sh.shard(data.val_mask)
    n_train = int(data.train_mask.sum())
    opt = torch.optim.SGD(base.parameters(), lr=0.1)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    errs = []
    for _ in range(3):
        ref.eval()
        with torch.no_grad():
            lp = ref(data)
        acc_ref = (lp[data.val_mask].argmax(1) == data.y[data.val_mask]).float().mean().item()   # of the weights as they stand
        ref.train()
        ropt.zero_grad()
        torch.nn.functional.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
        ropt.step()
        stats = sh.train_eval_step(opt, xl, yl, tl, vl, n_train)
        acc = (stats[0] / stats[1]).item()
        err_w = max((a - b).abs().max().item() for a, b in zip(base.parameters(), ref.parameters()))
        errs.append((abs(acc - acc_ref), err_w))
    sh.train()
    lp_tr, lp_ev = sh.forward_pair(xl)
    sh.eval()
    with torch.no_grad():
        e_ev = (lp_ev - sh(xl)).abs().max().item()
    ret[rank] = (max(e[0] for e in errs), max(e[1] for e in errs), e_ev)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_data_parallel_one_pass_epoch_gloo(world):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_pair_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, (dacc, err_w, e_ev) in ret.items():
        assert dacc < 1e-6 and err_w < TOL and e_ev < TOL, (rank, dacc, err_w, e_ev)


@pytest.mark.parametrize('world', [2, 3])
def test_data_parallel_ranks_gloo(world):
    """world 2 and 3 (61 nodes: blocks of 31 + 30 and of 21 + 21 + 19)"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, (err_fwd, err_w, dacc) in ret.items():
        assert err_fwd < TOL and err_w < TOL and dacc < 1e-6, (rank, err_fwd, err_w, dacc)


def _dp_chunked_worker(rank, world, port, ret):
    """``DCR_DP_CHUNKED_GATHER=1`` (models/gcn_dp.py: the [Z_train | Z_eval] exchange as two asynchronous all-gathers, the
    second under the first half's aggregation) against the one-shot exchange: the same bits in both outputs of
    forward_pair — all rows and selected rows — and the same weights after two train_eval_step epochs."""
    import copy
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models import gcn as gcn_mod
    from models.gcn import GCN
    from models.gcn_dp import ShardedGCN
    gcn_mod.set_aggregate_backend('torch')
    ei_np, n = synthetic.powerlaw_graph(1501, 4, seed=23)                  # 1501: uneven blocks at 2 and at 8 ranks
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 12, generator=g)
    y = torch.randint(0, 4, (n,), generator=g)
    r = torch.rand(n, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei_np), y=y, num_nodes=n, train_mask=r < 0.3, val_mask=(r >= 0.3) & (r < 0.6))
    torch.manual_seed(9)
    start = GCN(Dataset(data, 4), hidden=[16, 8], dropout=0.0)             # two exchanges per pass

    def run(chunked):
        os.environ['DCR_DP_CHUNKED_GATHER'] = '1' if chunked else '0'
        model = copy.deepcopy(start)
        sh = ShardedGCN(model, data.edge_index, n)
        xl, yl, tl, vl = sh.shard(data.x), sh.shard(data.y), sh.shard(data.train_mask), sh.shard(data.val_mask)
        sh.train()
        with torch.no_grad():
            all_rows = sh.forward_pair(xl)
            s_tr, s_ev = sh.row_selection(tl.nonzero().squeeze(1)), sh.row_selection(vl.nonzero().squeeze(1))
            some_rows = sh.forward_pair(xl, rows_train=s_tr, rows_eval=s_ev)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        stats = [sh.train_eval_step(opt, xl, yl, tl, vl, int(data.train_mask.sum())).clone() for _ in range(2)]
        return all_rows, some_rows, [p.detach().clone() for p in model.parameters()], stats
    real, n_async = dist.all_gather_into_tensor, [0]

    def counting(*args, **kw):
        n_async[0] += bool(kw.get('async_op'))
        return real(*args, **kw)
    dist.all_gather_into_tensor = counting
    a = run(False)
    assert n_async[0] == 0
    b = run(True)
    assert n_async[0] == 2 * 2 * 4                                          # two halves x two exchanges x (two forwards + two epochs)
    dist.all_gather_into_tensor = real
    os.environ.pop('DCR_DP_CHUNKED_GATHER')
    same = (all(torch.equal(u, v) for u, v in zip(a[0], b[0])) and all(torch.equal(u, v) for u, v in zip(a[1], b[1]))
            and all(torch.equal(u, v) for u, v in zip(a[2], b[2])) and all(torch.equal(u, v) for u, v in zip(a[3], b[3])))
    ret[rank] = (bool(same), int(a[1][0].shape[0]))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_data_parallel_chunked_exchange_gloo(world):
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_chunked_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    assert all(same for same, _ in ret.values()), dict(ret)
    assert sum(rows for _, rows in ret.values()) > 0


def _dp_world8_worker(rank, world, port, ret):
    """One of eight gloo ranks on a 20k-node power-law graph: degree-dealt partition, block-streamed set-up, logits of the
    sharded model against the single-process model (models/gcn.py:32-44) and one training step's weights."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models import gcn as gcn_mod
    from models.gcn import GCN
    from models.gcn_dp import ShardedGCN, balanced_partition
    gcn_mod.set_aggregate_backend('torch')
    ei_np, n = synthetic.powerlaw_graph(20000, 5, seed=17)
    ei = torch.from_numpy(ei_np)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, 24, generator=g)
    y = torch.randint(0, 5, (n,), generator=g)
    r = torch.rand(n, generator=g)
    data = Data(x=x, edge_index=ei, y=y, num_nodes=n, train_mask=r < 0.3, val_mask=(r >= 0.3) & (r < 0.6))
    torch.manual_seed(4)
    ref = GCN(Dataset(data, 5), hidden=[16], dropout=0.0)
    torch.manual_seed(4)
    base = GCN(Dataset(data, 5), hidden=[16], dropout=0.0)
    sh = ShardedGCN(base, ei, n)
    owner, index, per = balanced_partition(ei, n, world)
    deg = torch.bincount(ei[0], minlength=n)
    nnz_share = float((deg[owner == rank] + 1).sum()) / float((deg + 1).sum()) * world     # (+1: the self loop of every node)
    xl, yl, tl = sh.shard(data.x), sh.shard(data.y), sh.shard(data.train_mask)
    ref.eval(), sh.eval()
    with torch.no_grad():
        want = ref(data).index_select(0, sh.owned)
        got = sh(xl)[:sh.owned.numel()]
    err_fwd = (got - want).abs().max().item()
    ropt = torch.optim.Adam(ref.parameters(), lr=0.01)
    opt = torch.optim.Adam(base.parameters(), lr=0.01)
    ref.train()
    ropt.zero_grad()
    torch.nn.functional.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
    ropt.step()
    sh.train_step(opt, xl, yl, tl, int(data.train_mask.sum()))
    err_w = max((a - b).abs().max().item() for a, b in zip(base.parameters(), ref.parameters()))
    tr = sh.setup_transient_bytes(24)
    ret[rank] = (err_fwd, err_w, nnz_share, tr['streamed'] / tr['gathered'])
    dist.destroy_process_group()


def test_data_parallel_eight_ranks_gloo():
    """Eight ranks (the width of BASELINE.json's configs[4]) on the CPU: every rank's share of the non-zeros within 2 % of
    1 / 8, logits within 1e-5 of the single-process model, weights after a step too; the set-up transient is a quarter of the
    gathered matrix."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_world8_worker, args=(8, _free_port(), ret), nprocs=8, join=True)
    assert len(ret) == 8
    for rank, (err_fwd, err_w, share, transient) in ret.items():
        assert err_fwd < TOL and err_w < TOL, (rank, err_fwd, err_w)
        assert 0.98 < share < 1.02, (rank, share)
        assert transient <= 0.26, (rank, transient)


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('feat', [1, 7, 16, 30, 64, 128, 130, 256])
def test_spmm_hip_vs_torch(feat):
    from models import gcn
    from dcr import synthetic
    ei, n = synthetic.powerlaw_graph(3000, 6, seed=2)
    ei = torch.from_numpy(ei).cuda()
    csr = gcn.gcn_norm_csr(ei, None, n)
    torch.manual_seed(0)
    B = torch.randn(n, feat, device='cuda')
    bias = torch.randn(feat, device='cuda')
    for b, relu in ((None, False), (bias, False), (bias, True)):
        got = gcn._spmm_hip(csr.rowptr, csr.col, csr.val, B, n, b, relu)
        want = gcn._spmm_torch(csr.rowptr, csr.col, csr.val, B.double(), n, None if b is None else b.double(), relu)
        assert (got.double() - want).abs().max().item() < TOL, (feat, relu)
    got_t = gcn._spmm_hip(csr.rowptr_t, csr.col_t, csr.val_t, B, n)
    assert (got_t - gcn._spmm_hip(csr.rowptr, csr.col, csr.val, B, n)).abs().max().item() < TOL  # Â is symmetric here


@pytest.mark.gpu
def test_gcn_gpu_forward_backward_vs_dense_reference():
    from models.gcn import GCN, dense_reference_logits
    data, ds = _toy(n=500, f=40, c=6)
    data = data.to('cuda')
    torch.manual_seed(1)
    m = GCN(ds, hidden=[32], dropout=0.5).cuda()
    m.eval()
    got = m(data)
    want = dense_reference_logits(m, data.x, data.edge_index, data.num_nodes)
    assert (got.double() - want).abs().max().item() < TOL
    torch.nn.functional.nll_loss(got[data.train_mask], data.y[data.train_mask]).backward()
    g_got = [p.grad.clone() for p in m.parameters()]
    m.zero_grad()
    w64 = dense_reference_logits(m, data.x, data.edge_index, data.num_nodes)
    torch.nn.functional.nll_loss(w64[data.train_mask], data.y[data.train_mask]).backward()
    for a, p in zip(g_got, m.parameters()):
        assert (a - p.grad).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_rewire_then_train_end_to_end_gpu():
    """The reference's flow (save_models.py:46,74-85): rewire -> GCN -> Adam with two parameter groups -> training_loop."""
    from experiment.training_loop import evaluate, training_loop
    from models.gcn import GCN
    from rewiring.rewire import rewire
    data, ds = _toy(n=400, f=24, c=5, seed=9)
    np.random.seed(0)
    data.edge_index = rewire(data, 'bfc', 10, 0.95, 163)
    data = data.to('cuda')
    torch.manual_seed(0)
    m = GCN(ds, hidden=[16], dropout=0.3).cuda()
    opt = torch.optim.Adam([{'params': m.non_reg_params, 'weight_decay': 0},
                            {'params': m.reg_params, 'weight_decay': 0.01}], lr=0.02)
    m = training_loop(m, opt, data, epochs=30, patience=10)
    r = evaluate(m, data, test=True)
    assert 0.0 <= r['test_acc'] <= 1.0 and r['val_acc'] > 0.15


@pytest.mark.gpu
@pytest.mark.parametrize('K,M,N', [(5000, 128, 256), (4099, 16, 128), (777, 64, 1433), (130, 7, 128), (9, 33, 70),
                                   (100000, 128, 256), (0, 16, 32)])
def test_atb_mfma_vs_torch(K, M, N):
    """dW = dZᵀ·X on the hand-written f32 MFMA kernel against torch's fp64 product (asymmetric data: a swapped row /
    column map in the accumulator write-out cannot pass)."""
    from models.gcn import atb_hip
    g = torch.Generator(device='cuda').manual_seed(K + M + N)
    a = torch.randn(K, M, device='cuda', generator=g)
    b = torch.randn(K, N, device='cuda', generator=g) + torch.arange(N, device='cuda') * 0.01
    got = atb_hip(a, b)
    want = (a.double().t() @ b.double())
    scale = (a.double().abs().t() @ b.double().abs()).clamp_min(1.0)
    assert got.shape == (M, N)
    assert ((got.double() - want).abs() / scale).max().item() < 2e-6
    assert torch.equal(got, atb_hip(a, b))          # fixed summation order: bit-reproducible


@pytest.mark.gpu
def test_linear_backward_uses_mfma_kernel_and_matches_autograd():
    from models.gcn import _Linear
    torch.manual_seed(0)
    lin = _Linear(96, 40).cuda()
    x = torch.randn(3000, 96, device='cuda', requires_grad=True)
    y = lin(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    gx, gw = x.grad.clone(), lin.weight.grad.clone()
    gw_ref = (w.double().t() @ x.detach().double()).float()
    assert torch.allclose(gx, w @ lin.weight.detach(), atol=1e-4)
    assert torch.allclose(gw, gw_ref, rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_fused_relu_dropout():
    """One-pass ReLU+dropout: survivors are positive inputs scaled by 1/(1-p), the keep rate is 1-p, the backward pass
    uses the same bits, a seed reproduces the mask, evaluation mode is plain ReLU."""
    from models.gcn import relu_dropout
    act, drop = torch.nn.ReLU(), torch.nn.Dropout(p=0.3)
    torch.manual_seed(5)
    x = torch.randn(3001, 129, device='cuda', requires_grad=True)   # odd sizes: exercises the tail
    drop.train()
    y = relu_dropout(x, act, drop)
    kept = y != 0
    pos = x.detach() > 0
    assert not (kept & ~pos).any()
    rate = kept.sum().item() / pos.sum().item()
    assert abs(rate - 0.7) < 0.01, rate
    assert torch.equal(y[kept], (x.detach() * (1.0 / (1.0 - 0.3)))[kept].float())
    y.sum().backward()
    want = torch.where(kept, torch.full_like(y, float(np.float32(1.0 / (1.0 - 0.3)))), torch.zeros_like(y))
    assert torch.equal(x.grad, want)
    drop.eval()
    assert torch.equal(relu_dropout(x.detach(), act, drop), torch.relu(x.detach()))
    # successive calls draw different masks
    drop.train()
    assert not torch.equal(relu_dropout(x.detach(), act, drop) != 0, kept)


def _dp_worker_gpu(rank, world, port, ret):
    """Two ranks on ONE MI355X (gloo moves the tensors; the kernels are the product's HIP kernels): the sharded model
    with row-partitioned CSR, pre-propagated first layer and fused ReLU/dropout off against the single-process model."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        probe = torch.ones(4, device='cuda')
        dist.all_reduce(probe)
        gathered = torch.empty(8, device='cuda')
        dist.all_gather_into_tensor(gathered, torch.ones(4, device='cuda'))
    except Exception as e:  # this torch build's gloo cannot move device tensors
        ret[rank] = ('unsupported', str(e))
        dist.destroy_process_group()
        return
    from models.gcn import GCN
    from models.gcn_dp import ShardedGCN
    from dcr import synthetic
    from dcr.data import Data, Dataset
    ei, n = synthetic.powerlaw_graph(3001, 4, seed=9)                      # 3001: uneven blocks, hubs above 96 nnz
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 48, generator=g)
    y = torch.randint(0, 5, (n,), generator=g)
    r = torch.rand(n, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei), y=y, num_nodes=n, train_mask=r < 0.3,
                val_mask=(r >= 0.3) & (r < 0.6)).to('cuda')
    ds = Dataset(data, 5)
    torch.manual_seed(7)
    base = GCN(ds, hidden=[24], dropout=0.0).cuda()
    ref = GCN(ds, hidden=[24], dropout=0.0).cuda()
    ref.load_state_dict(base.state_dict())
    sh = ShardedGCN(base, data.edge_index, n)
    xl, yl, tl = sh.shard(data.x), sh.shard(data.y), sh.shard(data.train_mask)
    n_train = int(data.train_mask.sum())
    ref.eval(); sh.eval()
    with torch.no_grad():
        err_fwd = (sh(xl)[:sh.owned.shape[0]] - ref(data)[sh.owned]).abs().max().item()
    opt = torch.optim.SGD(base.parameters(), lr=0.1)
    sh.train_step(opt, xl, yl, tl, n_train)
    ref.train()
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    ropt.zero_grad()
    torch.nn.functional.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
    ropt.step()
    err_w = max((a - b).abs().max().item() for a, b in zip(base.parameters(), ref.parameters()))
    acc = sh.eval_correct(xl, yl, sh.shard(data.val_mask))
    ref.eval()
    with torch.no_grad():
        lp = ref(data)
    acc_ref = (lp[data.val_mask].argmax(1) == data.y[data.val_mask]).float().mean().item()
    ret[rank] = (err_fwd, err_w, abs(acc - acc_ref))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_data_parallel_two_ranks_hip_kernels():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker_gpu, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2
    if any(v[0] == 'unsupported' for v in ret.values()):
        pytest.skip(f'gloo cannot move device tensors in this torch build: {dict(ret)}')
    for rank, (err_fwd, err_w, dacc) in ret.items():
        assert err_fwd < 1e-5 and err_w < 1e-5 and dacc < 1e-6, (rank, err_fwd, err_w, dacc)


@pytest.mark.gpu
def test_cora_shaped_logits_within_tolerance():
    """BASELINE.json's tolerance (logits within 1e-5 of the reference GCN) at Cora's shape: 2,485 nodes, 1,433 sparse
    row-normalised features, hidden 128, 7 classes — the first layer runs as (Â·X)·Wᵀ, the check is the dense fp64
    restatement of PyG's Â·(X·Wᵀ)."""
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models.gcn import GCN, dense_reference_logits
    ei, n = synthetic.powerlaw_graph(2485, 2, seed=1)
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(n, 1433, generator=g) < 0.0127).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
    data = Data(x=x, edge_index=torch.from_numpy(ei), y=torch.randint(0, 7, (n,), generator=g), num_nodes=n).to('cuda')
    torch.manual_seed(3)
    model = GCN(Dataset(data, 7), hidden=[128], dropout=0.3396).cuda()
    model.eval()
    with torch.no_grad():
        got = model(data)
        want = dense_reference_logits(model, data.x, data.edge_index, n)
    err = (got.double() - want).abs().max().item()
    assert err < 1e-5, err


def test_balanced_partition_deals_nodes_by_degree():
    """The data-parallel partition (models/gcn_dp.py::balanced_partition): every rank the same number of nodes and, on a
    preferential-attachment graph whose early ids are the hubs, the same share of the non-zeros within 10 %
    (contiguous blocks of ids: 2.7 x the ideal share on rank 0 of 8)."""
    from dcr import synthetic
    from models.gcn_dp import balanced_partition
    ei_np, n = synthetic.powerlaw_graph(20001, 10, seed=3)
    ei = torch.from_numpy(ei_np)
    deg = torch.bincount(ei[0], minlength=n)
    for world in (2, 3, 8):
        owner, index, per = balanced_partition(ei, n, world)
        assert per == (n + world - 1) // world
        new_id = owner * per + index
        assert torch.unique(new_id).numel() == n and int(new_id.max()) < world * per        # a permutation into the padded range
        nodes = torch.bincount(owner, minlength=world)
        assert int(nodes.max() - nodes.min()) <= 1
        nnz = torch.zeros(world, dtype=torch.int64).index_add_(0, owner, deg + 1)             # rows of A + I
        share = nnz.double() / nnz.sum() * world
        assert float(share.max()) < 1.1 and float(share.min()) > 0.9, share
        blocks = torch.zeros(world, dtype=torch.int64).index_add_(0, torch.arange(n) // per, deg + 1)
        assert float(blocks.max()) / float(blocks.sum()) * world > float(share.max())          # what contiguous blocks would give


@pytest.mark.gpu
@pytest.mark.parametrize('hidden,classes,n', [(128, 16, 5003), (64, 6, 2120), (128, 7, 65)])
def test_activation_fused_into_the_next_contraction(hidden, classes, n):
    """dcr_act_linear_fwd / _bwd (models/gcn.py:36-42 between two layers, one pass): the training operand equals the separate
    fused ReLU + dropout kernel followed by a float64 contraction (same Philox stream: same keep mask), the evaluation
    operand relu(x)·Wᵀ, the gradients those of the separate ops; pair, train-only and eval-only calls agree bit for bit."""
    from models import gcn
    from models.gcn import _ActLinearFn, _ReluDropoutFn
    gcn.set_aggregate_backend('hip')
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, hidden, device=dev, generator=g)
    w = (torch.randn(classes, hidden, device=dev, generator=g) * 0.1)
    p = 0.4
    ctr = gcn._dropout_counter(dev)
    c0 = ctr.clone()
    h_ref = _ReluDropoutFn.apply(x, p)                       # the separate kernel: mask of call number c0
    ctr.copy_(c0)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    z_tr, z_ev = _ActLinearFn.apply(xr, wr, p, True, True)
    want_tr = (h_ref.double() @ w.double().t()).float()
    want_ev = (torch.relu(x).double() @ w.double().t()).float()
    assert (z_tr - want_tr).abs().max().item() < 1e-4 and (z_ev - want_ev).abs().max().item() < 1e-4
    gz = torch.randn(n, classes, device=dev, generator=g)
    z_tr.backward(gz)
    keep = h_ref != 0
    want_gx = torch.where(keep, (gz.double() @ w.double()).float() / (1 - p), torch.zeros_like(x))
    assert (xr.grad - want_gx).abs().max().item() < 1e-4
    want_w = gz.double().t() @ h_ref.double()
    assert (wr.grad.double() - want_w).abs().max().item() <= max(1e-5, 4.0 * 1.1920929e-07 * float(n) ** 0.5) * want_w.abs().max().item()
    ctr.copy_(c0)
    only_tr = _ActLinearFn.apply(x, w, p, True, False)[0]
    only_ev = _ActLinearFn.apply(x, w, 0.0, False, True)[1]
    assert torch.equal(only_tr, z_tr.detach()) and torch.equal(only_ev, z_ev)
    # behind a layer with a bias: its bias gradient comes out of the same backward pass (column sums of dx)
    from models.gcn import _LinearFn
    a = torch.randn(n, 24, device=dev, generator=g)
    w1 = (torch.randn(hidden, 24, device=dev, generator=g) * 0.2).requires_grad_(True)
    b1 = torch.zeros(hidden, device=dev, requires_grad=True)
    before = _LinearFn.colsum_handoffs
    ctr.copy_(c0)
    pre = _LinearFn.apply(a, w1, b1)
    z, _ = _ActLinearFn.apply(pre, wr.detach(), p, True, False)
    z.backward(gz)
    assert _LinearFn.colsum_handoffs == before + 1
    ctr.copy_(c0)
    keep1 = _ReluDropoutFn.apply(pre.detach(), p) != 0          # the separate kernel again: the mask of call number c0
    want_gpre = torch.where(keep1, (gz.double() @ w.double()).float() / (1 - p), torch.zeros_like(pre)).double()
    rel = max(1e-5, 4.0 * 1.1920929e-07 * float(n) ** 0.5)   # float32 accumulation over n rows, relative to the largest entry
    want_b1, want_w1 = want_gpre.sum(0), want_gpre.t() @ a.double()
    assert (b1.grad.double() - want_b1).abs().max().item() <= rel * want_b1.abs().max().item()
    assert (w1.grad.double() - want_w1).abs().max().item() <= rel * want_w1.abs().max().item()


def _ReluDropoutKeep(gcn, pre, p, ctr, c0):
    """keep mask (ReLU AND dropout) of dropout call number c0 on this pre-activation, from the stand-alone kernel."""
    from models.gcn import _ReluDropoutFn
    saved = ctr.clone()
    ctr.copy_(c0)
    keep = _ReluDropoutFn.apply(pre.detach(), p) != 0
    ctr.copy_(saved)
    return keep


@pytest.mark.gpu
@pytest.mark.parametrize('feats,hidden,classes,n', [(256, 128, 16, 5003), (48, 64, 6, 2120), (16, 128, 7, 65), (32, 64, 16, 1),
                                                    (320, 64, 3, 40000),
                                                    # round 5, the K-chunked kernel: the reference's own widths (Cora, Citeseer:
                                                    # utils/hyperparams.py:2-21), a width that is no multiple of 4, a wide multiple
                                                    # of 16, and enough rows for a workgroup to walk several row groups
                                                    (1433, 128, 7, 2485), (3703, 64, 6, 2120), (23, 64, 6, 333), (512, 128, 16, 1000),
                                                    (1433, 128, 7, 70001)])
def test_first_layer_activation_and_next_lin_in_one_kernel(feats, hidden, classes, n, monkeypatch):
    """dcr_first_layer_fwd_ws_f32_dev (models/gcn.py:36-42 from the first GCNConv's x to the second GCNConv's lin, on Â·X), both
    kernels behind it — W1 resident in LDS, and W1 streamed through LDS in K chunks with the last workgroup of a row group summing
    the chunks' partial tiles: the pre-activation against a float64 contraction; both outputs and the keep bits EQUAL to what
    dcr_act_linear_fwd_f32_dev makes of that same pre-activation (same Philox stream, same order of operations); pair, train-only
    and eval-only calls agree bit for bit, and so do two calls in a row (the chunk sums do not depend on which workgroup arrives
    last); the gradients — from the one-kernel backward (dcr_first_layer_bwd_f32_dev) and from the separate kernels — against
    float64 contractions of the two-kernel route's d loss / d pre."""
    import ctypes
    from dcr import _lib
    from models import gcn
    from models.gcn import _ActLinearFn, _FirstLayerFn
    gcn.set_aggregate_backend('hip')
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(11)
    f16 = (feats + 15) // 16 * 16
    axp = torch.zeros(n, f16, device=dev)                      # Â·X as the kernels read it: width rounded up to 16, zero pad
    axp[:, :feats] = torch.randn(n, feats, device=dev, generator=g)
    ax = axp[:, :feats]
    w1 = torch.randn(hidden, feats, device=dev, generator=g) * (feats ** -0.5)
    b1 = torch.randn(hidden, device=dev, generator=g) * 0.1
    w2 = torch.randn(classes, hidden, device=dev, generator=g) * 0.1
    assert _lib.lib().dcr_first_layer_fits(feats, hidden, classes) == 1
    p = 0.4
    ctr = gcn._dropout_counter(dev)
    c0 = ctr.clone()
    # the kernel by its C entry point: pre, bits, both outputs
    words = ctypes.c_int64()
    _lib.check(_lib.lib().dcr_relu_dropout_bits_words(n * hidden, ctypes.byref(words)))
    bits = torch.zeros(words.value, dtype=torch.int64, device=dev)
    pre = torch.empty(n, hidden, device=dev)
    both = torch.empty(n, 2 * classes, device=dev)
    cur = torch.cuda.current_stream(dev).cuda_stream
    st = ctypes.c_void_p(cur)
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    ws = gcn._first_layer_workspace(dev, cur, n, feats, hidden)
    resident = feats % 16 == 0 and 4 * (hidden * ((feats + 63) // 64 * 64) + 17 * hidden) <= 160 * 1024
    assert (ws is None) == resident
    ws_ptr, ws_n = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())

    def forward(pre_t, both_t, bits_t, dwords=None):
        ctr.copy_(c0)
        _lib.check(_lib.lib().dcr_first_layer_fwd_ws_f32_dev(axp.data_ptr(), f16, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), pre_t.data_ptr(),
                                                             both_t.data_ptr(), both_t.data_ptr() + 4 * classes, 2 * classes, bits_t.data_ptr(),
                                                             None if dwords is None else dwords.data_ptr(),
                                                             n, feats, hidden, classes, p, seed, 0, ctr.data_ptr(), ws_ptr, ws_n, st))
    forward(pre, both, bits)
    # the dropout decisions drawn ahead of the call (dcr_dropout_words_dev, the epoch driver's side stream): the same bits
    dwords = torch.full_like(bits, -1)
    ctr.copy_(c0)
    _lib.check(_lib.lib().dcr_dropout_words_dev(dwords.data_ptr(), n, hidden, p, seed, 0, ctr.data_ptr(), st))
    pre_d, both_d, bits_d = torch.empty_like(pre), torch.empty_like(both), torch.zeros_like(bits)
    forward(pre_d, both_d, bits_d, dwords)
    assert torch.equal(pre_d, pre) and torch.equal(bits_d, bits) and torch.equal(both_d, both)
    assert torch.equal(bits & ~dwords, torch.zeros_like(bits))                     # bits = decisions AND (pre > 0)
    rc = _lib.lib().dcr_first_layer_fwd_ws_f32_dev(axp.data_ptr(), f16, w1.data_ptr(), None, w2.data_ptr(), None, None, both_d.data_ptr(), classes,
                                                   None, dwords.data_ptr(), n, feats, hidden, classes, 0.0, 0, 0, None, ws_ptr, ws_n, st)
    assert rc != 0                                                                 # decisions without a training output
    del pre_d, both_d, bits_d
    if ws is not None:
        n_groups = ((n + 15) // 16 + 3) // 4
        assert int(ws[-n_groups:].view(torch.int32).abs().sum().item()) == 0      # the tickets are zero again
        pre2, both2, bits2 = torch.empty_like(pre), torch.empty_like(both), torch.zeros_like(bits)
        forward(pre2, both2, bits2)                                               # the same bits whichever workgroup arrives last
        assert torch.equal(pre2, pre) and torch.equal(both2, both) and torch.equal(bits2, bits)
    want_pre = (ax.double() @ w1.double().t() + b1.double())
    assert (pre.double() - want_pre).abs().max().item() < 2e-5 * max(1.0, want_pre.abs().max().item())
    xr = pre.clone().requires_grad_(True)
    w2r = w2.clone().requires_grad_(True)
    ctr.copy_(c0)
    z_tr, z_ev = _ActLinearFn.apply(xr, w2r, p, True, True)                  # the two-kernel route on the SAME pre-activation
    assert torch.equal(both[:, :classes], z_tr.detach()) and torch.equal(both[:, classes:], z_ev)
    ctr.copy_(c0)
    w1f, b1f, w2f = (t.clone().requires_grad_(True) for t in (w1, b1, w2))
    f_tr, f_ev = _FirstLayerFn.apply(axp, w1f, b1f, w2f, p, True, True)
    assert torch.equal(f_tr.detach(), z_tr.detach()) and torch.equal(f_ev, z_ev) and not f_ev.requires_grad
    gz = torch.randn(n, classes, device=dev, generator=g)
    z_tr.backward(gz)
    gpre = xr.grad                                                            # d loss / d pre of the two-kernel route
    # Bounds that follow from float32 accumulation over n rows (round 5; before: 1e-2 * sqrt(n) ABSOLUTE, i.e. 2.7 % of a typical
    # entry at the bench shape — a dropped 16-row tail unit passed): a sum of n float32 products in a fixed order is within
    # ~ eps * sqrt(n) of the exact sum relative to the size of its terms' running total; 4 eps sqrt(n) (at least 1e-5) of the
    # largest reference entry.
    rel = max(1e-5, 4.0 * 1.1920929e-07 * float(n) ** 0.5)
    want_w1 = gpre.double().t() @ ax.double()
    want_b1 = gpre.double().sum(0)
    h_ref = torch.where(_ReluDropoutKeep(gcn, pre, p, ctr, c0), pre.double() / (1 - p), torch.zeros_like(pre, dtype=torch.float64))
    want_w2 = gz.double().t() @ h_ref

    def close(got, want):
        return (got.double() - want).abs().max().item() <= rel * max(want.abs().max().item(), 1e-30)
    for one_kernel in ('1', '0'):                                             # dcr_first_layer_bwd_f32_dev / the separate kernels
        monkeypatch.setenv('DCR_FIRST_BWD_FUSED', one_kernel)
        for t in (w1f, b1f, w2f):
            t.grad = None
        f_tr.backward(gz, retain_graph=True)
        assert w1f.grad.shape == w1.shape
        if one_kernel == '0':
            assert torch.equal(w2f.grad, w2r.grad)
        assert close(w2f.grad, want_w2) and close(w2r.grad, want_w2)
        assert close(b1f.grad, want_b1)
        assert close(w1f.grad, want_w1)
    if n > 64:
        # the bound has teeth: the one-kernel backward called on n - 16 rows (a dropped tail unit) FAILS the dW1 line
        short = n - 16
        need = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_first_layer_bwd_workspace(short, feats, hidden, ctypes.byref(need)))
        wsb = torch.empty(max(need.value, 1), device=dev)
        gw1, gb1, gw2 = torch.empty(hidden, feats, device=dev), torch.empty(hidden, device=dev), torch.empty(classes, hidden, device=dev)
        _lib.check(_lib.lib().dcr_first_layer_bwd_f32_dev(gz.data_ptr(), w2.data_ptr(), bits.data_ptr(), pre.data_ptr(), axp.data_ptr(), f16,
                                                          gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), wsb.data_ptr(), need.value, short,
                                                          feats, hidden, classes, p, st))
        assert not close(gw1, want_w1)
        assert close(gw1, gpre[:short].double().t() @ ax[:short].double())   # (and it is the right answer for those rows)
    ctr.copy_(c0)
    only_tr = _FirstLayerFn.apply(axp, w1, b1, w2, p, True, False)[0]
    only_ev = _FirstLayerFn.apply(axp, w1, b1, w2, 0.0, False, True)[1]
    assert torch.equal(only_tr, f_tr.detach()) and torch.equal(only_ev, f_ev)
    unpadded = _FirstLayerFn.apply(ax.contiguous(), w1, b1, w2, 0.0, False, True)[1]      # (a caller that did not pad)
    assert torch.equal(unpadded, f_ev)
    no_bias = _FirstLayerFn.apply(axp, w1, None, w2, 0.0, False, True)[1]
    want = (torch.relu(ax.double() @ w1.double().t()) @ w2.double().t())
    assert (no_bias.double() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    # shapes outside the kernels' reach are refused, not approximated; a width beyond W1's LDS image needs the workspace
    lib = _lib.lib()
    assert lib.dcr_first_layer_fits(3703, 64, 6) == 1 and lib.dcr_first_layer_fits(1433, 128, 7) == 1
    assert lib.dcr_first_layer_fits(256, 96, 16) == 0 and lib.dcr_first_layer_fits(256, 128, 17) == 0
    rc = lib.dcr_first_layer_fwd_ws_f32_dev(axp.data_ptr(), f16 - 4, w1.data_ptr(), None, w2.data_ptr(), None, None, both.data_ptr(),
                                            classes, None, None, n, feats, hidden, classes, 0.0, 0, 0, None, ws_ptr, ws_n, st)
    assert rc != 0                                                             # row stride below the padded width
    if ws is not None:
        rc = lib.dcr_first_layer_fwd_f32_dev(axp.data_ptr(), f16, w1.data_ptr(), None, w2.data_ptr(), None, None, both.data_ptr(),
                                             classes, None, n, feats, hidden, classes, 0.0, 0, 0, None, st)
        assert rc != 0                                                         # the round-4 entry point has no workspace to give


@pytest.mark.gpu
def test_model_with_the_one_kernel_first_layer_equals_the_separate_kernels(monkeypatch):
    """GCN.forward / forward_pair on a shape the one-kernel first layer takes (32 features, hidden 64): logits and gradients
    within float32 rounding of the route through the GEMM library + dcr_act_linear_fwd_f32_dev (DCR_FIRST_FUSED=0), and of the
    dense float64 restatement; training, evaluation and the one-pass epoch see the same numbers."""
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models import gcn
    from models.gcn import GCN, dense_reference_logits
    gcn.set_aggregate_backend('hip')
    dev = torch.device('cuda', 0)
    ei, n = synthetic.powerlaw_graph(3000, 4, seed=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 32, generator=g)
    y = torch.randint(0, 5, (n,), generator=g)
    data = Data(x=x.to(dev), edge_index=torch.from_numpy(ei).to(dev), y=y.to(dev), num_nodes=n)
    torch.manual_seed(0)
    model = GCN(Dataset(data, 5), hidden=[64], dropout=0.5).to(dev)
    with torch.no_grad():
        model.layers[0].bias.uniform_(-0.1, 0.1)
    calls = {'n': 0}
    real = gcn._FirstLayerFn.apply

    def counting(*a):
        calls['n'] += 1
        return real(*a)
    monkeypatch.setattr(gcn._FirstLayerFn, 'apply', staticmethod(counting))
    ctr = gcn._dropout_counter(dev)
    start = ctr.clone()

    def run(fused):
        monkeypatch.setenv('DCR_FIRST_FUSED', '1' if fused else '0')
        ctr.copy_(start)                                  # the same dropout masks in both runs
        model.zero_grad()
        model.eval()
        with torch.no_grad():
            ev = model(data)
        model.train()
        c0 = ctr.clone()
        tr = model(data)
        loss = torch.nn.functional.nll_loss(tr, data.y)
        loss.backward()
        grads = [q.grad.clone() for q in model.parameters()]
        ctr.copy_(c0)
        p_tr, p_ev = model.forward_pair(data)
        ctr.copy_(c0)
        rows = torch.arange(0, n, 7, device=dev)
        r_tr, r_ev = model.forward_pair(data, rows_train=rows, rows_eval=rows)
        return ev, tr.detach(), grads, p_tr.detach(), p_ev, r_tr.detach(), r_ev, rows

    before = calls['n']
    ev1, tr1, g1, ptr1, pev1, rtr1, rev1, rows = run(True)
    assert calls['n'] == before + 4                       # eval, train, pair, pair with rows
    ev0, tr0, g0, ptr0, pev0, _, _, _ = run(False)
    assert calls['n'] == before + 4
    assert torch.equal(ptr1, tr1) and torch.equal(pev1, ev1)
    assert torch.equal(rtr1, tr1[rows]) and torch.equal(rev1, ev1[rows])
    assert (ev1 - ev0).abs().max().item() < 1e-5
    # (a pre-activation within rounding of zero may land on the other side of the ReLU: a handful of elements out of 192,000)
    assert (tr1 - tr0).abs().max().item() < 1e-4
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())
    model.eval()
    want = dense_reference_logits(model, data.x, data.edge_index, n)
    assert (ev1.double() - want).abs().max().item() < 1e-5
