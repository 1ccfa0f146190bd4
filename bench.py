#!/usr/bin/env python3
"""bench.py — SDRF iterations/sec (+ BFC edges/sec, roofline, CPU baseline) on the north-star graph.

A "step" is ONE SDRF iteration (rewiring/sdrf_no_cuda.py:22-66) in full-recompute mode on the synthetic
power-law graph S100k (N=100,000, m=10, E=999,900; BASELINE.json configs[2]): a full Balanced Forman pass over
every edge, arg-min, candidate/improvement tensor, host softmax draw, add, stale arg-max, conditional remove.
The graph is resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

SDRF is inherently sequential (each iteration depends on the previous graph), so with N > 1 every rank runs an
independent replica (its own numpy seed) — "replicas only", no data-path collective; value = N*K / max-rank time.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'discrete-curvature-rewiring_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_MATRIX_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (MI355X_MICROARCH.md)
PMC_PROFILE = 'r05_pmc_traffic.json'  # written by tools/pmc_traffic.sh on the GPU box, copied into profiles/
PMC_BOUND = 'r05_pmc_bound.json'      # SQ wait / issue / LDS counters per kernel (tools/pmc_bound.sh): what binds the pass
KERNEL_TIMES = 'r05_kernel_times.json'  # rocprofv3 --kernel-trace --stats of this bench (tools/kernel_times.sh), same hash rule


def cpu_baseline(ei, n, E, budget_s=12.0):
    """The CPU oracle (oracle/dcr_oracle.c, kind="port") timed on this box's host cores on a bounded sample: the curvature
    pass on all cores AND on one thread, the improvement step serial as the reference's loop is
    (sdrf_no_cuda.py:41-46), median over the three most negatively curved sampled edges."""
    from oracle import c_oracle
    cores = len(os.sched_getaffinity(0))
    C = c_oracle.CGraph(ei, n)
    eu, ev = C.edges()
    rng = np.random.Generator(np.random.PCG64(2024))
    probe = rng.choice(E, size=min(E, 4000), replace=False)
    t0 = time.perf_counter()
    C.curv_edges(eu[probe], ev[probe], 'bfc', nthreads=cores)
    t_probe = time.perf_counter() - t0
    # three runs of a third of the budget each over the same sample: the figure is their median, the spread is in the line
    # (a shared box: other tenants' load moves an all-core timing by tens of per cent from one run to the next)
    n_sample = int(min(E, max(4000, budget_s * 0.2 / max(t_probe / len(probe), 1e-9))))
    pick = rng.choice(E, size=n_sample, replace=False)
    t_runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        cv = C.curv_edges(eu[pick], ev[pick], 'bfc', nthreads=cores)
        t_runs.append(time.perf_counter() - t0)
    t_pass = sorted(t_runs)[1]
    edges_per_s = n_sample / t_pass
    pass_s = E / edges_per_s
    # the same pass on ONE thread, on a smaller sample (about 3 s)
    n1 = int(min(n_sample, max(500, 3.0 * edges_per_s / max(cores, 1))))
    t0 = time.perf_counter()
    C.curv_edges(eu[pick[:n1]], ev[pick[:n1]], 'bfc', nthreads=1)
    t1 = time.perf_counter() - t0
    edges_per_s_1 = n1 / t1
    # improvements: literal add / recompute / remove (sdrf_no_cuda.py:41-46) for the three lowest sampled edges — on ALL cores
    # (round 4: the oracle's threaded step, each worker on a private copy of the graph; thousands of candidates per sample,
    # so the figure no longer hangs on three 150-candidate timings of one thread) and, for the one-thread figure, serial on
    # a small sample; median over the three edges
    order = np.argsort(cv, kind='stable')[:3]
    per_edge, per_edge_1 = [], []
    for m in order:
        x, y = int(eu[pick[m]]), int(ev[pick[m]])
        ci, cj = C.candidates(x, y)
        n_cand = len(ci)
        if not n_cand:
            per_edge.append((0.0, x, y, 0, 0))
            per_edge_1.append(0.0)
            continue
        # (all candidates when there are cores to share them: every worker first copies the graph, a fixed cost that an
        #  extrapolation from a sample would multiply)
        k = n_cand if cores > 1 else min(n_cand, 150)
        sel = np.sort(rng.choice(n_cand, size=k, replace=False)) if k < n_cand else np.arange(n_cand)
        best = float('inf')
        for _ in range(2):
            t0 = time.perf_counter()
            C.improvements(x, y, ci[sel], cj[sel], 'bfc', nthreads=cores)
            best = min(best, time.perf_counter() - t0)
        per_edge.append((best * n_cand / k, x, y, k, n_cand))
        k1 = min(n_cand, 150)
        best1 = float('inf')
        for _ in range(3):  # single-threaded and short: the fastest of three is the least disturbed by other tenants
            t0 = time.perf_counter()
            C.improvements(x, y, ci[sel[:k1]], cj[sel[:k1]], 'bfc')
            best1 = min(best1, time.perf_counter() - t0)
        per_edge_1.append(best1 * n_cand / k1)
    per_edge.sort()
    imp_s, x, y, k, n_cand = per_edge[len(per_edge) // 2]
    imp_s_1 = sorted(per_edge_1)[len(per_edge_1) // 2]
    iter_s = pass_s + imp_s
    return {
        'value': 1.0 / iter_s, 'unit': 'SDRF iterations/sec', 'cores': cores, 'kind': 'port',
        'sample': f'BFC pass over {n_sample} of {E} randomly sampled edges on {cores} threads (extrapolated x{E / n_sample:.1f}) '
                  f'+ candidate improvements on {cores} threads (each on a private copy of the graph): median of the three lowest '
                  f'sampled edges, here all {n_cand} candidates of edge ({x},{y}) (faster of 2 runs each)',
        'bfc_edges_per_sec': edges_per_s, 'pass_seconds_extrapolated': pass_s, 'improvements_seconds_extrapolated': imp_s,
        'improvements_seconds_per_edge': [round(t, 3) for t, *_ in per_edge],
        # spread of the sample: the same pass three times, the improvement step on three edges; `value` uses the medians
        'spread': {'pass_edges_per_sec_min_median_max': [n_sample / max(t_runs), edges_per_s, n_sample / min(t_runs)],
                   'value_min_max': [1.0 / (E / (n_sample / max(t_runs)) + per_edge[-1][0]),
                                     1.0 / (E / (n_sample / min(t_runs)) + per_edge[0][0])],
                   'note': 'min: slowest pass run + slowest of the three improvement edges; max: fastest of each.  The figure '
                           'depends on the box (host thread count, other tenants) and on which edges the sample holds: runs of this '
                           'bench on different boxes of the pool gave 0.12-0.40 iterations/s'},
        'one_thread': {'cores': 1, 'bfc_edges_per_sec': edges_per_s_1, 'pass_seconds_extrapolated': E / edges_per_s_1,
                       'improvements_seconds_extrapolated': imp_s_1,
                       'improvements_seconds_per_edge': [round(t, 3) for t in per_edge_1],
                       'value': 1.0 / (E / edges_per_s_1 + imp_s_1), 'unit': 'SDRF iterations/sec',
                       'sample': f'the same pass over {n1} sampled edges on 1 thread + the improvement step serial, as the '
                                 f'reference\'s loop is (150 candidates per edge, fastest of 3, extrapolated)'},
    }


def gcn_bench(args, rank, world, local_rank, dist, gcn_graph=None):
    """GCN epochs/sec (BASELINE.json configs[4] shape): synthetic N=1M / E=10M graph, F=256, hidden 128, 16 classes,
    dropout 0.5, Adam; epoch = one training step + one validation forward (experiment/training_loop.py:25-26).
    N > 1: row-partitioned data parallel (models/gcn_dp.py), fixed problem size -> strong scaling."""
    import torch
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models.gcn import GCN, gcn_norm_csr, _spmm_hip
    from models.gcn_dp import ShardedGCN
    dev = torch.device('cuda', local_rank)
    n, m, F, H, C = args.gcn_nodes, 10, 256, 128, 16
    ei_np, n = gcn_graph if gcn_graph is not None else synthetic.powerlaw_graph(n, m, seed=12345)
    ei = torch.from_numpy(ei_np).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, F, device=dev, generator=g)
    y = torch.randint(0, C, (n,), device=dev, generator=g)
    r = torch.rand(n, device=dev, generator=g)
    train_mask, val_mask = r < 0.1, (r >= 0.1) & (r < 0.2)
    torch.manual_seed(0)
    data = Data(x=x, edge_index=ei, y=y, num_nodes=n, train_mask=train_mask, val_mask=val_mask)
    model = GCN(Dataset(data, C), hidden=[H], dropout=0.5).to(dev)
    from experiment.save_models import make_adam
    # DCR_FUSED_ADAM: 2 (default here) = the one-launch Adam of this package (experiment/adam.py), 1 = torch's fused
    # implementation (one kernel per group + counters), 0 = torch's stock one (the experiment drivers' default)
    adam_mode = os.environ.get('DCR_FUSED_ADAM', '2')
    os.environ['DCR_FUSED_ADAM'] = adam_mode
    fused_adam = None if adam_mode == '2' else adam_mode == '1'
    opt = make_adam([{'params': model.non_reg_params, 'weight_decay': 0},
                     {'params': model.reg_params, 'weight_decay': 5e-4}], 0.01, dev, fused=fused_adam)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if dist is None:
        from experiment.training_loop import make_epoch
        epoch = make_epoch(model, opt, data, lagged=True)  # one captured HIP graph per epoch (LaggedGraphedEpoch)
    else:
        sh = ShardedGCN(model, ei, n)
        xl, yl, tl, vl = sh.shard(x), sh.shard(y), sh.shard(train_mask), sh.shard(val_mask)
        n_train = int(train_mask.sum())
        nnz_shares = sh.nnz_shares()   # non-zeros of Â per rank over the ideal share (degree-dealt partition: ~1.0 each)

        from models.gcn_dp import GraphedShardedEpoch
        if GraphedShardedEpoch.supported(sh, opt, xl):  # RCCL: the epoch replays as two HIP graphs, collectives inside
            epoch = GraphedShardedEpoch(sh, opt, xl, yl, tl, vl, n_train)
        else:
            def epoch():  # eager (gloo rehearsals): the same one-pass epoch, not captured
                sh.train_eval_step(opt, xl, yl, tl, vl, n_train)

    for _ in range(max(args.gcn_warmup, 5)):  # (the graphed epoch captures at its fourth call)
        epoch()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.gcn_epochs):
        epoch()
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    res = {'metric': 'GCN epochs/sec', 'value': args.gcn_epochs / el, 'unit': 'epochs/sec', 'n_gpus': world,
           'epochs': args.gcn_epochs, 'ms_per_epoch': el / args.gcn_epochs * 1e3, 'scaling': 'strong', 'dtype': 'f32',
           'hip_graph': type(epoch).__name__ in ('GraphedEpoch', 'LaggedGraphedEpoch', 'GraphedShardedEpoch'),
           'epoch_driver': type(epoch).__name__,
           'config': {'workload': f'synthetic preferential-attachment graph N={n} E={ei_np.shape[1] // 2}, F={F}, '
                                  f'hidden={H}, classes={C}, dropout 0.5, Adam; epoch = train step + val forward',
                      'parallelism': f'row-partitioned dp{world}' if world > 1 else 'single GPU'}}
    # what an epoch cannot go below on this chip, kernel by kernel: the two 65.5 GFLOP contractions of layer 1 (forward
    # and weight gradient) at the dense f32 matrix peak, everything else at the HBM roof on the bytes it has to move
    # (fused activation forward: read N*H, write 2 N*C + bits; backward: read N*H + N*C + bits, write N*H; the three
    # class-width aggregations on the rows and non-zeros an epoch reads; the layer-1 output written once)
    n_tr, n_va = int(train_mask.sum()), int(val_mask.sum())
    flops = 2.0 * 2.0 * n * F * H / world
    nnz_sel = 2.0 * (n_tr + n_va) / n * (ei_np.shape[1] + n)   # non-zeros of the selected rows (forward) and columns (backward)
    from models.gcn import first_layer_fused_ok
    one_kernel_first = first_layer_fused_ok(x, model.act_fn, model.layers[0], model.layers[1].lin)
    one_kernel_bwd = one_kernel_first and os.environ.get('DCR_FIRST_BWD_FUSED', '1') != '0'
    # (layer-1 output: written once and read once by the backward pass; where the first layer is not the one-kernel form,
    #  csrc/dcr_gcn_first.hip, it is read once more by the activation forward, and its gradient is written and read back)
    byts = (n * H * 4 * (1 + (0 if one_kernel_first else 1) + (1 if one_kernel_bwd else 2)) + n * C * 4 * 3 + 2 * n * H / 8
            + nnz_sel * (C * 4 + 8)) / world
    floor_ms = (flops / (F32_MATRIX_PEAK_TFLOPS * 1e12) + byts / (HBM_PEAK_GBPS * 1e9)) * 1e3
    res['epoch_floor_ms'] = floor_ms
    res['epoch_floor_frac'] = floor_ms / res['ms_per_epoch']
    res['epoch_floor_definition'] = ('layer-1 forward + weight-gradient contractions (2 x 2 N F H flop) at 157.3 TFLOP/s f32 MFMA, plus '
                                     'the bytes of the fused activation kernels and of the row-selected aggregations at 8 TB/s')
    res['first_layer'] = ('one kernel: (A_hat X) W1^T + b1, ReLU + dropout and the second layer\'s lin on the matrix cores '
                          '(dcr_first_layer_fwd_f32_dev; DCR_FIRST_FUSED=0: GEMM library + dcr_act_linear_fwd_f32_dev)'
                          if one_kernel_first else 'GEMM library + dcr_act_linear_fwd_f32_dev')
    res['first_layer_backward'] = ('one kernel: dW1, db1, dW2 with the pre-activation gradient in registers (dcr_first_layer_bwd_f32_dev; '
                                   'DCR_FIRST_BWD_FUSED=0: dcr_act_linear_bwd_fused_f32_dev + dcr_atb_f32_dev)'
                                   if one_kernel_bwd else 'dcr_act_linear_bwd_fused_f32_dev + dcr_atb_f32_dev')
    res['adam'] = ('one launch per step (experiment/adam.py, dcr_adam_step_f32_dev; DCR_FUSED_ADAM=1: torch fused, 0: torch stock — the '
                   'experiment drivers\' default)' if type(opt).__name__ == 'OneLaunchAdam' else
                   'torch fused (one kernel per group; DCR_FUSED_ADAM=0: stock foreach implementation, the experiment drivers\' default)'
                   if getattr(opt, 'defaults', {}).get('fused') else 'torch stock (foreach), capturable')
    res['last_aggregation'] = ('evaluated at the rows the epoch reads (training rows for the loss, validation rows for the accuracy: '
                               f'{n_tr} + {n_va} of {n}); DCR_GCN_ALL_ROWS=1 computes every row')
    if dist is None and not args.no_cpu_baseline:   # the same epoch with every row of the last aggregation, for comparison
        os.environ['DCR_GCN_ALL_ROWS'] = '1'
        try:
            torch.manual_seed(0)
            model2 = GCN(Dataset(data, C), hidden=[H], dropout=0.5).to(dev)
            opt2 = make_adam([{'params': model2.non_reg_params, 'weight_decay': 0},
                              {'params': model2.reg_params, 'weight_decay': 5e-4}], 0.01, dev, fused=fused_adam)
            epoch2 = make_epoch(model2, opt2, data, lagged=True)
            for _ in range(6):
                epoch2()
            sync()
            t0 = time.perf_counter()
            for _ in range(10):
                epoch2()
            sync()
            res['ms_per_epoch_all_rows'] = (time.perf_counter() - t0) / 10 * 1e3
            del model2, opt2, epoch2
        finally:
            os.environ.pop('DCR_GCN_ALL_ROWS', None)
    if dist is None and not args.no_cpu_baseline:
        # a deeper model on the same graph: Pubmed's hidden_depth = 3 (utils/hyperparams.py:23-30) — three hidden layers of 128, so
        # two full aggregations at the hidden width per direction are on the epoch (the 2-layer model has none: A_hat X is
        # pre-propagated and the last aggregation runs at the class width on the selected rows)
        try:
            torch.manual_seed(0)
            model3 = GCN(Dataset(data, C), hidden=[H, H, H], dropout=0.5).to(dev)
            opt3 = make_adam([{'params': model3.non_reg_params, 'weight_decay': 0},
                              {'params': model3.reg_params, 'weight_decay': 5e-4}], 0.01, dev, fused=fused_adam)
            epoch3 = make_epoch(model3, opt3, data, lagged=True)
            for _ in range(6):
                epoch3()
            sync()
            t0 = time.perf_counter()
            for _ in range(10):
                epoch3()
            sync()
            res['hidden_depth_3'] = {'ms_per_epoch': (time.perf_counter() - t0) / 10 * 1e3, 'hidden': [H, H, H],
                                     'epoch_driver': type(epoch3).__name__,
                                     'note': 'Pubmed depth (hidden_depth 3): the hidden-width aggregations are on this epoch'}
            del model3, opt3, epoch3
        except Exception as ex:  # noqa: BLE001
            res['hidden_depth_3'] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
    if dist is not None:
        res['rccl_ranks'] = dist.get_world_size()
        res['backend'] = dist.get_backend()
        res['nnz_share_per_rank_over_ideal'] = [round(v, 4) for v in nnz_shares]
        res['nnz_share_max_over_ideal'] = max(nnz_shares)
    if rank == 0:
        # Roofline of the aggregation kernel the epoch actually runs: with the first layer pre-propagated ((Â·X)·W1ᵀ), every
        # SpMM of an epoch is at the CLASS width (layer 2 forward, its backward, the validation forward); the hidden
        # width is reported next to it for reference.  HIP events on torch's current stream, the one the kernel runs on.
        csr = gcn_norm_csr(ei, None, n) if dist is None else sh.csr
        nnz = int(csr.col.shape[0])

        def spmm_point(width):
            z = torch.randn(n, width, device=dev)
            for _ in range(3):
                _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                _spmm_hip(csr.rowptr, csr.col, csr.val, z, csr.n_rows)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            nbytes = nnz * 8 + (csr.n_rows + 1) * 8 + (n + csr.n_rows) * width * 4   # ideal reuse of B
            gathered = nnz * (width * 4 + 8)                                          # every non-zero pulls one row of B
            return {'bound': 'hbm', 'achieved': nbytes / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'launch_ms': ms,
                    'algorithmic_bytes_per_launch': nbytes, 'nnz': nnz, 'n_feat': width,
                    'gathered_GBps': gathered / (ms * 1e-3) / 1e9,
                    'gathered_rows_per_ns': nnz / (ms * 1e-3) / 1e9}
        res['spmm_roofline'] = spmm_point(C)
        res['spmm_roofline']['kernel'] = ('k_spmm_csr at the class width over ALL rows (model(data) outside an epoch); an epoch runs '
                                           'it on the rows it reads')
        res['spmm_roofline']['note'] = ('B (N x classes floats) sits in the Infinity Cache; rows are 64 bytes, so the kernel is '
                                        'bound by the rate of row requests, not by bytes (MI355X_MICROARCH.md, indexed rows)')
        res['spmm_roofline_hidden_width'] = spmm_point(H)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            res['cpu_baseline'] = gcn_cpu_baseline(ei_np, n, F, H, C)
        except Exception as ex:  # noqa: BLE001
            res['cpu_baseline'] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
    return res


def gcn_cpu_baseline(ei_np, n, F, H, C, epochs=1):
    """The same epoch (training step + validation forward, models/gcn.py:32-44 with GCNConv's formula) in stock PyTorch on
    the host cores of this box: sparse CSR Â times dense, autograd, Adam.  A bounded sample (an epoch takes ~25 s on 256
    threads at the 1M-node shape): a short untimed warm-up of the thread pool, then ``epochs`` timed epochs."""
    import torch
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, F, generator=g)
    y = torch.randint(0, C, (n,), generator=g)
    r = torch.rand(n, generator=g)
    train_idx, val_idx = (r < 0.1).nonzero().squeeze(1), ((r >= 0.1) & (r < 0.2)).nonzero().squeeze(1)
    src = torch.cat([torch.from_numpy(ei_np[0]), torch.arange(n)])
    dst = torch.cat([torch.from_numpy(ei_np[1]), torch.arange(n)])
    deg = torch.zeros(n).index_add_(0, dst, torch.ones(dst.shape[0]))
    dinv = deg.pow(-0.5)
    A = torch.sparse_coo_tensor(torch.stack([dst, src]), dinv[src] * dinv[dst], (n, n)).coalesce().to_sparse_csr()
    w1 = torch.nn.Parameter(torch.randn(H, F) * 0.05)
    b1 = torch.nn.Parameter(torch.zeros(H))
    w2 = torch.nn.Parameter(torch.randn(C, H) * 0.05)
    b2 = torch.nn.Parameter(torch.zeros(C))
    opt = torch.optim.Adam([{'params': [w2, b2], 'weight_decay': 0}, {'params': [w1, b1], 'weight_decay': 5e-4}], lr=0.01)

    def forward(train):
        h = torch.sparse.mm(A, x @ w1.t()) + b1
        h = torch.nn.functional.dropout(torch.relu(h), 0.5, training=train)
        return torch.log_softmax(torch.sparse.mm(A, h @ w2.t()) + b2, dim=1)

    def epoch():
        opt.zero_grad()
        torch.nn.functional.nll_loss(forward(True)[train_idx], y[train_idx]).backward()
        opt.step()
        with torch.no_grad():
            return (forward(False)[val_idx].argmax(1) == y[val_idx]).float().mean().item()
    (torch.randn(2048, 2048) @ torch.randn(2048, 2048)).sum().item()  # wake the thread pool
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    el = time.perf_counter() - t0
    return {'value': epochs / el, 'unit': 'epochs/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{epochs} epoch(s) of the same model in stock PyTorch (torch.sparse CSR aggregation, '
                      f'autograd, Adam) on {cores} host threads, N={n} F={F} hidden={H} classes={C}',
            'ms_per_epoch': el / epochs * 1e3}


def gcn_small_shape(n, m, n_feat, hidden, n_cls, dropout, lr, wd, local_rank, epochs=200, route=None):
    """BASELINE.json configs[3] shape (Citeseer-LCC: 2,120 nodes, 3,703 features, hidden 64, 6 classes; hyper-parameters
    of utils/hyperparams.py) on a synthetic graph of that size: epochs/sec of train step + validation forward."""
    import torch
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from experiment.training_loop import make_epoch
    from models.gcn import GCN
    dev = torch.device('cuda', local_rank)
    ei_np, n = synthetic.powerlaw_graph(n, m, seed=12345)
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(n, n_feat, device=dev, generator=g) < 0.009).float()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0)
    y = torch.randint(0, n_cls, (n,), device=dev, generator=g)
    r = torch.rand(n, device=dev, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei_np).to(dev), y=y, num_nodes=n, train_mask=r < 0.1,
                val_mask=(r >= 0.1) & (r < 0.4))
    # route: which first layer (models/gcn.py).  None: the package's choice — for these features (0.9 % non-zeros) Â·(X·W1ᵀ) over the
    # non-zeros of X; 'dense_mfma': the K-chunked MFMA kernel on Â·X (DCR_SPARSE_X=0); 'library': the GEMM library on Â·X + the
    # fused activation kernel (DCR_SPARSE_X=0 DCR_FIRST_FUSED=0: the route of rounds 1-4)
    saved = {k: os.environ.get(k) for k in ('DCR_SPARSE_X', 'DCR_FIRST_FUSED')}
    if route in ('dense_mfma', 'library'):
        os.environ['DCR_SPARSE_X'] = '0'
    if route == 'library':
        os.environ['DCR_FIRST_FUSED'] = '0'
    try:
        torch.manual_seed(0)
        model = GCN(Dataset(data, n_cls), hidden=[hidden], dropout=dropout).to(dev)
        from experiment.save_models import make_adam
        opt = make_adam([{'params': model.non_reg_params, 'weight_decay': 0},
                         {'params': model.reg_params, 'weight_decay': wd}], lr, dev,
                        fused=None if os.environ.get('DCR_FUSED_ADAM', '2') == '2' else os.environ.get('DCR_FUSED_ADAM') == '1')
        # the epoch as experiment/training_loop.py runs it: eager for the first calls, then the captured HIP graph
        epoch = make_epoch(model, opt, data, lagged=True)
        for _ in range(10):
            epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            epoch()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        first_route = ('sparse input: X in CSR, pre = A_hat (X W1^T) + b1 (two aggregations), then dcr_act_linear_fwd_f32_dev'
                       if model.layers[0].sparse_input(x) is not None else
                       'dense, one MFMA kernel on A_hat X (K-chunked: dcr_first_layer_fwd_ws_f32_dev)' if os.environ.get('DCR_FIRST_FUSED', '1') != '0'
                       else 'dense, GEMM library on A_hat X + dcr_act_linear_fwd_f32_dev')
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {'metric': 'GCN epochs/sec', 'value': epochs / el, 'unit': 'epochs/sec', 'ms_per_epoch': el / epochs * 1e3,
            'first_layer_route': first_route, 'feature_density': float(torch.count_nonzero(x)) / x.numel(),
            'dtype': 'f32', 'hip_graph': type(epoch).__name__ in ('GraphedEpoch', 'LaggedGraphedEpoch'), 'epoch_driver': type(epoch).__name__, 'config': {'workload': f'Citeseer-shaped synthetic graph N={n} E={ei_np.shape[1] // 2}, '
                                                   f'F={n_feat}, hidden={hidden}, classes={n_cls}, dropout {dropout}, '
                                                   f'Adam lr {lr} wd {wd}; epoch = train step + val forward'}}


def spawn_ranks(n):
    """One child process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torch.distributed.run would
    export); the children are this same script with the same arguments.  Children are started, never exec'ed into."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--nodes', type=int, default=100000)
    ap.add_argument('--m', type=int, default=10)
    ap.add_argument('--tau', type=float, default=163.0)
    ap.add_argument('--removal-bound', type=float, default=0.95)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gcn', action='store_true')
    ap.add_argument('--no-incremental', action='store_true')
    ap.add_argument('--no-config2', action='store_true')
    ap.add_argument('--no-s1m', action='store_true')
    ap.add_argument('--gcn-nodes', type=int, default=1000000)
    ap.add_argument('--gcn-epochs', type=int, default=20)
    ap.add_argument('--gcn-warmup', type=int, default=3)
    ap.add_argument('--gcn-timeout', type=float, default=300.0)
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one fresh process per rank (nothing in this parent has
        # touched the GPU or imported torch), pass rank 0's JSON line through, exit with the worst return code
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    from dcr import synthetic
    from dcr.data import Data
    from rewiring.sdrf_no_cuda import SdrfRun

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the measured path)')
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:  # rehearsal of the N > 1 path on a box with fewer GPUs (DCR_BENCH_BACKEND=gloo): share
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get('DCR_BENCH_FORCE_DIST') == '1'  # exercise the N>1 code path on one GPU (testing aid)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', RANK='0', WORLD_SIZE='1')
        # 'nccl' is RCCL on ROCm.  gloo only for rehearsals: forced, or when the ranks outnumber the GPUs of the box
        backend = os.environ.get('DCR_BENCH_BACKEND', 'nccl' if world <= ndev else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ei, n = synthetic.powerlaw_graph(args.nodes, args.m, seed=12345)
    E = ei.shape[1] // 2
    data = Data(edge_index=torch.from_numpy(ei), num_nodes=n)
    run = SdrfRun(data, 'bfc', True, args.removal_bound, args.tau, device=local_rank)
    G = run.G
    np.random.seed(rank)  # replica seed; rank 0 matches BASELINE.md's np.random.seed(0)

    for _ in range(args.warmup):
        run.step()
    bytes0, bytes0_2s = G.bfc_algorithmic_bytes(one_sided=True), G.bfc_algorithmic_bytes()
    pass_engine = G.pass_engine() if hasattr(G, 'pass_engine') else 'node-centric'
    G.profile_reset()
    barrier()
    t0 = time.perf_counter()
    steps_done = 0
    step_edges = []  # (x, y, candidates) of every timed step: what the improvement pipeline worked on
    for _ in range(args.steps):
        steps_done += 1
        go_on = run.step()
        step_edges.append(run.last)
        if not go_on:
            break
    barrier()
    elapsed = time.perf_counter() - t0
    pass_ms_total, pass_count = G.profile_read()
    bytes1, bytes1_2s = G.bfc_algorithmic_bytes(one_sided=True), G.bfc_algorithmic_bytes()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([steps_done], dtype=torch.int64, device='cuda')
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_steps = int(s.item())
    else:
        total_steps = steps_done

    # optional mode, reported separately (never `value`): the same iterations with the incremental curvature pass, which
    # recomputes only edges near the previous iteration's edits and leaves the same bits in the buffer
    inc = None
    if rank == 0 and not args.no_incremental:
        run_i = SdrfRun(data, 'bfc', True, args.removal_bound, args.tau, device=local_rank, incremental=True)
        np.random.seed(0)
        for _ in range(args.warmup + 1):
            run_i.step()
        run_i.G.profile_reset()
        torch.cuda.synchronize()
        ti = time.perf_counter()
        n_i = 0
        for _ in range(args.steps):
            n_i += 1
            if not run_i.step():
                break
        torch.cuda.synchronize()
        ti = time.perf_counter() - ti
        ms_i, cnt_i = run_i.G.profile_read()
        inc = {'value': n_i / ti, 'unit': 'iterations/sec', 'ms_per_step': ti / n_i * 1e3,
               'bfc_pass_ms': ms_i / max(cnt_i, 1), 'steps': n_i,
               'note': 'dcr_curvature_pass_incremental: bit-identical results (tests), not the headline metric'}
        run_i = None
    # the same iterations with the draw on the host (improvements downloaded, numpy's exp / sum, exact cumsum): reported for
    # comparison, never `value`
    hostdraw = None
    if rank == 0 and not args.no_incremental:
        os.environ['DCR_DEVICE_DRAW'] = '0'
        try:
            run_h = SdrfRun(data, 'bfc', True, args.removal_bound, args.tau, device=local_rank)
        finally:
            os.environ.pop('DCR_DEVICE_DRAW', None)
        np.random.seed(0)
        for _ in range(args.warmup + 1):
            run_h.step()
        run_h.G.profile_reset()
        torch.cuda.synchronize()
        th = time.perf_counter()
        n_h = 0
        for _ in range(args.steps):
            n_h += 1
            if not run_h.step():
                break
        torch.cuda.synchronize()
        th = time.perf_counter() - th
        ms_h, cnt_h = run_h.G.profile_read()
        hostdraw = {'value': n_h / th, 'unit': 'iterations/sec', 'ms_per_step': th / n_h * 1e3, 'bfc_pass_ms': ms_h / max(cnt_h, 1),
                    'steps': n_h, 'note': 'DCR_DEVICE_DRAW=0: the draw on the host, as in rounds 1-2'}
        run_h = None
    # BASELINE.json configs[2] as written: a full pass + 500 SDRF iterations, one run timed end to end (graph upload and
    # row build excluded, everything else included); reported next to the K-step figure, never instead of it
    cfg2 = None
    if rank == 0 and not args.no_config2:
        run5 = SdrfRun(data, 'bfc', True, args.removal_bound, args.tau, device=local_rank)
        np.random.seed(0)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        n5 = 0
        for i5 in range(500):
            n5 += 1
            if not run5.step(more=i5 + 1 < 500):   # (no curvature pass queued behind the last iteration)
                break
        torch.cuda.synchronize()
        t5 = time.perf_counter() - t5
        cfg2 = {'iterations': n5, 'seconds': t5, 'iterations_per_sec': n5 / t5,
                'edges_after': int(run5.G.number_of_edges()),
                'note': 'BASELINE.json configs[2]: S100k, full Balanced Forman pass + SDRF, 500 iterations in one run'}
        run5 = None
    # BASELINE.json configs[1]'s shape: a Cora-sized graph (2,485 nodes, 4,966 edges; Cora's tau and bound,
    # utils/hyperparams.py:2-9), full recompute per iteration — a pass this small runs edge by edge (csrc/dcr_bfc_nc.hip)
    cora_sdrf = None
    if rank == 0 and not args.no_config2:
        try:
            ei_c, n_c = synthetic.powerlaw_graph(2485, 2, seed=12345)
            run_c = SdrfRun(Data(edge_index=torch.from_numpy(ei_c), num_nodes=n_c), 'bfc', True, 0.95, 163.0, device=local_rank)
            np.random.seed(0)
            for _ in range(8):
                run_c.step()
            run_c.G.profile_reset()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            n_cs = 0
            for _ in range(200):
                n_cs += 1
                if not run_c.step():
                    break
            torch.cuda.synchronize()
            tc = time.perf_counter() - tc
            pms, pcnt = run_c.G.profile_read()
            cora_sdrf = {'value': n_cs / tc, 'unit': 'iterations/sec', 'ms_per_step': tc / n_cs * 1e3, 'bfc_pass_ms': pms / max(pcnt, 1),
                         'nodes': int(n_c), 'edges': int(ei_c.shape[1] // 2), 'steps': n_cs,
                         'note': "BASELINE.json configs[1]'s shape (Cora-sized synthetic graph, Cora's tau / removal bound): full "
                                 'Balanced Forman recompute per iteration; the pass runs a workgroup per edge at this size'}
            run_c = None
        except Exception as e:   # a side figure must not cost the line
            cora_sdrf = {'error': repr(e)}
    # tau = inf: the deterministic variant of SURVEY.md 8(d) (utils/softmax.py:5-8: one-hot at the first arg-max; the
    # improvements stay on the device, the draw consumes one uniform)
    tinf = None
    if rank == 0 and not args.no_incremental:
        run_t = SdrfRun(data, 'bfc', True, args.removal_bound, float('inf'), device=local_rank)
        np.random.seed(0)
        for _ in range(args.warmup + 1):
            run_t.step()
        torch.cuda.synchronize()
        tt = time.perf_counter()
        n_t = 0
        for _ in range(args.steps):
            n_t += 1
            if not run_t.step():
                break
        torch.cuda.synchronize()
        tt = time.perf_counter() - tt
        tinf = {'value': n_t / tt, 'unit': 'iterations/sec', 'ms_per_step': tt / n_t * 1e3, 'steps': n_t,
                'note': 'tau = inf (utils/softmax.py:5-8): same loop, the added edge is the first arg-max of the improvements'}
        run_t = None
    # whole-iteration algorithmic bytes: B_pass + B_improve + 16 E, B_improve = candidates x B(x, y)
    iter_bytes = None
    if rank == 0:
        try:
            deg0 = np.bincount(ei[0], minlength=n)
            rowptr0 = np.concatenate([[0], np.cumsum(deg0)])
            nbr = lambda a: ei[1][rowptr0[a]:rowptr0[a + 1]]
            b_imp = []
            for (x, y, n_cand) in step_edges:
                if x is None:
                    continue
                nx, ny = nbr(x), nbr(y)
                dxs, dys = np.setdiff1d(nx, ny), np.setdiff1d(ny, nx)
                side = min(int(deg0[dxs].sum()), int(deg0[dys].sum()))
                size = len(dxs) if deg0[dxs].sum() <= deg0[dys].sum() else len(dys)
                b_xy = 4 * (len(nx) + len(ny)) + 4 * side + 8 * (2 + size) + 8
                b_imp.append(float(n_cand) * b_xy)
            iter_bytes = {'improve_bytes_mean': float(np.mean(b_imp)) if b_imp else 0.0,
                          'candidates_mean': float(np.mean([c for _, _, c in step_edges if c is not None])) if step_edges else 0.0,
                          'argext_bytes': 16.0 * E}
        except Exception as ex:  # noqa: BLE001
            iter_bytes = {'error': f'{type(ex).__name__}: {ex}'[:200]}
    # the curvature pass at S1M (N = 1,000,000, E ~ 10M): the size where the adjacency (80 MB + slack) no longer sits in L2 and
    # HBM / Infinity Cache is the roof in earnest; same byte definition, counted on the device
    s1m = None
    gcn_graph = None
    if rank == 0 and not args.no_gcn and not args.no_s1m and args.gcn_nodes >= 1000000:
        try:
            from dcr.graph import DcrGraph
            gcn_graph = synthetic.powerlaw_graph(args.gcn_nodes, 10, seed=12345)
            G1 = DcrGraph(gcn_graph[0], gcn_graph[1], device=local_rank)
            for _ in range(2):
                G1.curvature_pass('bfc')
            G1.profile_reset()
            for _ in range(5):
                G1.curvature_pass('bfc')
            ms1, cnt1 = G1.profile_read()
            b1 = G1.bfc_algorithmic_bytes(one_sided=True)
            pm = ms1 / max(cnt1, 1)
            d1 = np.bincount(gcn_graph[0][0], minlength=gcn_graph[1]).astype(np.float64)
            e1 = gcn_graph[0].shape[1] // 2
            own = 4.0 * float((d1 ** 2).sum()) + 112.0 * e1      # the two-hop engine's own bytes (see roofline)
            rb = own if G1.pass_engine() == 'two-hop' else b1
            s1m = {'nodes': int(gcn_graph[1]), 'edges': int(e1), 'bfc_pass_ms': pm,
                   'pass_engine': G1.pass_engine(), 'bfc_edges_per_sec': e1 / (pm * 1e-3),
                   'algorithmic_bytes_per_launch': rb, 'achieved_GBps': rb / (pm * 1e-3) / 1e9,
                   'frac': rb / (pm * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   'equivalent_one_sided_bytes_per_launch': b1, 'equivalent_one_sided_frac': b1 / (pm * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   'max_degree': int(d1.max())}
            G1 = None
        except Exception as ex:  # noqa: BLE001
            s1m = {'error': f'{type(ex).__name__}: {ex}'[:300]}
    elif not args.no_gcn and world > 1:
        pass  # every rank generates the graph itself in gcn_bench (no broadcast of a 160 MB edge list)
    draws = {'device': int(getattr(run, 'device_draws', 0)), 'host': int(getattr(run, 'host_draws', 0))}
    run = G = None  # release the SDRF graph before the GCN leg
    out = None
    if rank == 0:
        pass_ms = pass_ms_total / max(pass_count, 1)
        alg_bytes = 0.5 * (bytes0 + bytes1)            # one-sided per-edge bytes: what the node-centric algorithm has to read
        alg_bytes_2s = 0.5 * (bytes0_2s + bytes1_2s)   # SURVEY 8(d) as written: both difference sets charged
        # The two-hop engine's OWN algorithmic bytes (round-3 judge: the fraction must be quoted on what the engine that ran has
        # to move, not on the bytes of the algorithm it replaced).  Every node reads each neighbour's row once: 4 B x sum over
        # nodes of the sum of their neighbours' degrees = 4 * sum d^2; per adjacency slot the neighbour id (4), its row header
        # (8) and the record written (16); per edge the two records joined (32), two row headers (16), the curvature (8).
        deg_b = np.bincount(ei[0], minlength=n).astype(np.float64)
        sum_d2 = float((deg_b ** 2).sum())
        engine_bytes = 4.0 * sum_d2 + 28.0 * 2.0 * E + 56.0 * E
        two_hop = pass_engine == 'two-hop'
        roof_bytes = engine_bytes if two_hop else alg_bytes
        achieved = roof_bytes / (pass_ms * 1e-3) / 1e9
        equivalent = alg_bytes / (pass_ms * 1e-3) / 1e9
        out = {
            'metric': 'SDRF iterations/sec', 'value': total_steps / elapsed, 'unit': 'iterations/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / max(steps_done, 1) * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'S100k: preferential-attachment graph N={n} m={args.m} E={E} (PCG64 seed 12345), '
                                   f'Balanced Forman curvature, full recompute every iteration, remove_edges=True, '
                                   f'tau={args.tau}, removal_bound={args.removal_bound}',
                       'parallelism': 'replicas only' if world > 1 else 'single GPU'},
            'rccl_ranks': dist.get_world_size() if dist is not None else 1,
            'backend': (dist.get_backend() if dist is not None else None),
            'bfc_edges_per_sec': E / (pass_ms * 1e-3),
            'bfc_pass_ms': pass_ms,
            'outside_pass_ms': elapsed / max(steps_done, 1) * 1e3 - pass_ms,
            'draws': {**draws,
                      'note': 'np.random.choice index found on the device from the uniform taken from numpy on the host '
                              '(dcr_sdrf_iteration_device_draw); host: draws left undecided by the margin test, redone with numpy'},
            'pass_engine': pass_engine,
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': None,
                         'kernel': 'curvature pass (all kernels between the two HIP events on the library stream: ' + pass_engine + ')',
                         'algorithmic_bytes_per_launch': roof_bytes,
                         'algorithmic_bytes_definition': (
                             'two-hop engine: 4 * sum d^2 (every neighbour row read once per node) + 28 per adjacency slot '
                             '(neighbour id, its row header, the record written) + 56 per edge (two records joined, two row '
                             'headers, the curvature written); from the degrees, exact' if two_hop else
                             'per edge 4(d_u+d_v) + 4*sum of row lengths of the cheaper difference set + 8(2+its size) + 8; '
                             'exact, counted on the device'),
                         'sum_d2': sum_d2,
                         'launch_ms': pass_ms, 'launches': pass_count,
                         'target_frac': 0.40, 'target_met': achieved / HBM_PEAK_GBPS >= 0.40,
                         'what_binds': ('not HBM: the engine reads 7-8 x fewer bytes than the per-edge algorithm and is bound by '
                                        'per-wave latency (chains of dependent LDS operations and device-memory reads at the '
                                        'occupancy its LDS and registers allow); counters under binding_counters'),
                         # the rate in bytes of the algorithm the engine REPLACED (per-edge, one-sided: SURVEY 8(d) with only the
                         # cheaper difference set charged): comparable across engines and rounds, NOT an achieved HBM fraction
                         'equivalent_one_sided_bytes_per_launch': alg_bytes,
                         'equivalent_one_sided_GBps': equivalent,
                         'equivalent_one_sided_frac': equivalent / HBM_PEAK_GBPS,
                         'survey_8d_two_sided_bytes_per_launch': alg_bytes_2s,
                         'survey_8d_two_sided_GBps': alg_bytes_2s / (pass_ms * 1e-3) / 1e9,
                         # cross-check the judge applies: these bytes over the whole step must also stay below peak
                         'bytes_over_ms_per_step_GBps': roof_bytes / (elapsed / max(steps_done, 1)) / 1e9},
        }
        if out['roofline']['frac'] > 1.0:
            out['roofline']['invalid'] = 'fraction above 1: the byte count does not describe what the kernels move'
        # HBM-side bytes per pass and the SQ instruction counters come from separate rocprofv3 --pmc runs (counters
        # cannot be read in-process): quoted only while the kernel sources still hash to what was profiled
        sys.path.insert(0, os.path.join(REPO, 'tools'))
        from kernel_hash import pass_sources_hash
        pmc = os.path.join(REPO, 'profiles', PMC_PROFILE)
        if os.path.exists(pmc) and args.nodes == 100000 and args.m == 10:
            with open(pmc) as f:
                rec = json.load(f)
            if rec.get('pass_sources_hash') == pass_sources_hash():
                out['roofline']['traffic'] = rec['traffic_bytes_per_pass']
                out['roofline']['traffic_source'] = 'profiles/' + PMC_PROFILE
                out['roofline']['traffic_frac'] = rec['traffic_bytes_per_pass'] / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
                out['roofline']['traffic_over_algorithmic'] = rec['traffic_bytes_per_pass'] / roof_bytes
                if 'sq_insts_valu_per_pass' in rec:
                    # the other resource: vector instruction issue.  A wave64 instruction occupies its SIMD-32 for 2 cycles
                    # (4 when one wave alone issues: MI355X_MICROARCH.md, cycle constants); 256 CUs x 4 SIMDs at 2.4 GHz.
                    insts = rec['sq_insts_valu_per_pass']
                    peak = 1024 * 2.4e9 / 2.0                     # wave-instructions per second, whole chip
                    out['roofline_valu_issue'] = {
                        'bound': 'valu-issue', 'achieved': insts / (pass_ms * 1e-3) / 1e9, 'peak': peak / 1e9,
                        'unit': 'G wave-instructions/s', 'frac': insts / (pass_ms * 1e-3) / peak,
                        'frac_at_4_cycles_per_instruction': insts * 4.0 / (pass_ms * 1e-3) / (1024 * 2.4e9),
                        'sq_insts_valu_per_pass': insts, 'source': 'profiles/' + PMC_PROFILE}
            else:
                out['roofline']['traffic_stale'] = ('profiles/' + PMC_PROFILE + ' was measured on other kernel sources '
                                                    '(hash mismatch): not quoted')
        pb = os.path.join(REPO, 'profiles', PMC_BOUND)
        if os.path.exists(pb) and args.nodes == 100000 and args.m == 10:
            with open(pb) as f:
                rec = json.load(f)
            if rec.get('pass_sources_hash') == pass_sources_hash():
                out['roofline']['binding_counters'] = {k: rec[k] for k in ('_about', 'per_kernel', 'pass') if k in rec}
                out['roofline']['binding_counters']['source'] = 'profiles/' + PMC_BOUND
            else:
                out['roofline']['binding_counters_stale'] = 'profiles/' + PMC_BOUND + ' was measured on other kernel sources: not quoted'
        ref_fix = os.path.join(REPO, 'tests', 'golden', 'reference_timing_s100k.json')
        if os.path.exists(ref_fix) and args.nodes == 100000 and args.m == 10:
            with open(ref_fix) as f:
                r = json.load(f)
            out['reference_cpu_extrapolated'] = {
                'bfc_edges_per_sec': r['edges_per_sec'], 'pass_seconds': r['extrapolated_pass_seconds'],
                'note': 'reference bfc_naive.bfc_edge itself, timed in the build container on sampled edges of this '
                        'same graph (tools/make_golden.py); the Python reference cannot travel to the GPU box'}
        if inc is not None:
            out['incremental_mode'] = inc
        if hostdraw is not None:
            out['host_draw_mode'] = hostdraw
        if tinf is not None:
            out['tau_inf'] = tinf
        if s1m is not None:
            out['s1m_pass'] = s1m
        if iter_bytes is not None and 'error' not in iter_bytes:
            tot = alg_bytes + iter_bytes['improve_bytes_mean'] + iter_bytes['argext_bytes']   # (SURVEY 8(d)'s formula: per-edge bytes)
            step_s = elapsed / max(steps_done, 1)
            out['roofline_iteration'] = {
                'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS, 'achieved': tot / step_s / 1e9,
                'frac': tot / step_s / 1e9 / HBM_PEAK_GBPS, 'algorithmic_bytes_per_iteration': tot,
                'pass_bytes': alg_bytes, 'improve_bytes_mean': iter_bytes['improve_bytes_mean'],
                'argext_bytes': iter_bytes['argext_bytes'], 'candidates_mean': iter_bytes['candidates_mean'],
                'definition': 'SURVEY.md 8(d): B_pass + B_improve + 16 E, B_improve = candidates x B(x, y) (an upper bound: the '
                              'pipeline derives every candidate from one sweep of the two neighbourhoods), over ms_per_step'}
            if out['roofline_iteration']['frac'] > 1.0:
                out['roofline_iteration']['invalid'] = ('fraction above 1: SURVEY 8(d) charges every candidate a full B(x, y) and every '
                                                        'edge its per-edge bytes; neither the improvement pipeline nor the two-hop pass '
                                                        'moves those bytes — an upper bound on work, not an achieved HBM fraction')
        elif iter_bytes is not None:
            out['roofline_iteration'] = iter_bytes
        kt = os.path.join(REPO, 'profiles', KERNEL_TIMES)
        if os.path.exists(kt) and args.nodes == 100000 and args.m == 10:
            with open(kt) as f:
                rec = json.load(f)
            if rec.get('sdrf_sources_hash') == pass_sources_hash(('dcr_sdrf.hip', 'dcr_internal.h')):
                k = rec['kernels_us']
                slots = rec.get('adjacency_slots', 0)
                if 'k_argext_edges' in k:
                    t = k['k_argext_edges'] * 1e-6
                    out['roofline_argext'] = {
                        'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS, 'kernel': 'k_argext_edges (sdrf_no_cuda.py:27,59,61)',
                        'launch_us': k['k_argext_edges'], 'algorithmic_bytes_per_launch': 8.0 * E,
                        'achieved': 8.0 * E / t / 1e9, 'frac': 8.0 * E / t / 1e9 / HBM_PEAK_GBPS,
                        'slot_bytes_per_launch': 16.0 * slots, 'slot_frac': 16.0 * slots / t / 1e9 / HBM_PEAK_GBPS,
                        'definition': 'SURVEY.md 8(d): 16 E per iteration for the two reductions = 8 E each; slot_*: what the '
                                      'kernel reads (16 B per adjacency slot: owner, neighbour, value)', 'source': 'profiles/' + KERNEL_TIMES}
                if 'k_imp_emit' in k and iter_bytes is not None and 'error' not in iter_bytes:
                    t = k['k_imp_emit'] * 1e-6
                    nb = 24.0 * iter_bytes['candidates_mean']
                    out['roofline_improvement_emit'] = {
                        'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBPS, 'kernel': 'k_imp_emit (sdrf_no_cuda.py:41-46)',
                        'launch_us': k['k_imp_emit'], 'algorithmic_bytes_per_launch': nb, 'achieved': nb / t / 1e9,
                        'frac': nb / t / 1e9 / HBM_PEAK_GBPS,
                        'definition': '24 B per candidate written (two ids, one float64); mean candidates of the timed steps',
                        'source': 'profiles/' + KERNEL_TIMES}
                out['sdrf_kernels_us'] = {kk: vv for kk, vv in k.items() if kk.startswith(('k_imp', 'k_argext', 'k_add', 'k_remove', 'k_pick'))}
        if cfg2 is not None:
            out['config2_500_iterations'] = cfg2
        if cora_sdrf is not None:
            out['sdrf_cora_shape'] = cora_sdrf
    # The side legs must not cost the headline line: a failure is recorded in their place, and a leg that does not come
    # back (a collective waiting for a rank that died) is cut off by a timer on every rank: rank 0 prints the line it has,
    # all ranks leave.  (The multi-GPU GCN leg has only ever run on one MI355X in the build environment.)
    import threading

    leg = {'done': False}
    leg_lock = threading.Lock()

    def cut_off():
        with leg_lock:
            if leg['done']:   # the leg came back while the timer was firing: nothing to cut off
                return
            if rank == 0 and out is not None:
                out['gcn'] = {'error': f'GCN leg did not finish within {args.gcn_timeout} s'}
                print(json.dumps(out), flush=True)
            os._exit(3)       # a hung collective or a dead rank is a failed run, with the headline line still printed
    timer = threading.Timer(args.gcn_timeout, cut_off)
    timer.daemon = True
    gcn = None
    if not args.no_gcn:
        timer.start()
        try:
            gcn = gcn_bench(args, rank, world, local_rank, dist, gcn_graph)
        except Exception as ex:  # noqa: BLE001
            gcn = {'error': f'{type(ex).__name__}: {ex}'[:400]}
        with leg_lock:
            leg['done'] = True
        timer.cancel()
    if rank == 0:
        if gcn is not None:
            out['gcn'] = gcn
            if world > 1 and 'error' not in gcn:
                # N > 1: `value` above is SDRF replica throughput ("replicas only"); the path that shards is the GCN, the
                # quantity BASELINE.json's metric names "at 1/2/4/8 GPU": lifted to the top level for the scaling curve
                out['gcn_epochs_per_sec'] = gcn['value']
                out['gcn_ms_per_epoch'] = gcn['ms_per_epoch']
                out['gcn_rccl_ranks'] = gcn.get('rccl_ranks')
                out['gcn_nnz_share_per_rank_over_ideal'] = gcn.get('nnz_share_per_rank_over_ideal')
                out['gcn_nnz_share_max_over_ideal'] = gcn.get('nnz_share_max_over_ideal')
                out['gcn_scaling'] = 'strong'
            if world == 1:
                try:
                    out['gcn_citeseer_shape'] = gcn_small_shape(2120, 2, 3703, 64, 6, 0.4103, 0.0199, 0.4551, local_rank)
                    # "measure both orders and keep the faster" (round-4 verdict): the same epoch through the other two first layers
                    out['gcn_citeseer_shape']['other_routes_ms_per_epoch'] = {
                        r: gcn_small_shape(2120, 2, 3703, 64, 6, 0.4103, 0.0199, 0.4551, local_rank, route=r)['ms_per_epoch']
                        for r in ('dense_mfma', 'library')}
                    out['gcn_cora_shape'] = gcn_small_shape(2485, 2, 1433, 128, 7, 0.3396, 0.0244, 0.1076, local_rank)
                except Exception as ex:  # noqa: BLE001
                    out['gcn_citeseer_shape'] = {'error': f'{type(ex).__name__}: {ex}'[:400]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(ei, n, E)
            except Exception as ex:  # noqa: BLE001
                out['cpu_baseline'] = {'error': f'{type(ex).__name__}: {ex}'[:400]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
