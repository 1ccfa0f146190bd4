"""One RCCL rank of tests/test_gcn_dp_nccl_gpu.py (started as a fresh process per rank, never imported by pytest):
the data-parallel epoch replayed as a captured HIP graph with the collectives inside (models/gcn_dp.py::GraphedShardedEpoch)
must leave the weights and accuracies of the same epochs run eagerly (ShardedGCN.train_eval_step)."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'discrete-curvature-rewiring_amd')]


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', device_id=dev)
    from dcr import synthetic
    from dcr.data import Data, Dataset
    from models.gcn import GCN
    from models.gcn_dp import GraphedShardedEpoch, ShardedGCN
    ei_np, n = synthetic.powerlaw_graph(20001, 6, seed=5)      # 20001: padded blocks; hubs above the long-row threshold
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, 64, generator=g)
    y = torch.randint(0, 7, (n,), generator=g)
    r = torch.rand(n, generator=g)
    data = Data(x=x, edge_index=torch.from_numpy(ei_np), y=y, num_nodes=n, train_mask=r < 0.2,
                val_mask=(r >= 0.2) & (r < 0.5)).to(dev)
    n_train = int(data.train_mask.sum())

    def build():
        torch.manual_seed(3)
        model = GCN(Dataset(data, 7), hidden=[32], dropout=0.0).to(dev)   # no dropout: the two runs must agree bit for bit
        opt = torch.optim.Adam([{'params': model.non_reg_params, 'weight_decay': 0},
                                {'params': model.reg_params, 'weight_decay': 5e-4}], lr=0.01, capturable=True)
        sh = ShardedGCN(model, data.edge_index, n)
        return model, opt, sh, (sh.shard(data.x), sh.shard(data.y), sh.shard(data.train_mask), sh.shard(data.val_mask))

    epochs = 9
    m_e, o_e, sh_e, (xl, yl, tl, vl) = build()
    acc_e = []
    for _ in range(epochs):
        st = sh_e.train_eval_step(o_e, xl, yl, tl, vl, n_train)
        acc_e.append((st[0] / st[1].clamp(min=1)).item())
    m_g, o_g, sh_g, (xg, yg, tg, vg) = build()
    assert GraphedShardedEpoch.supported(sh_g, o_g, xg), 'RCCL backend with a capturable optimiser: the epoch must capture'
    ep = GraphedShardedEpoch(sh_g, o_g, xg, yg, tg, vg, n_train)
    acc_g = [ep() for _ in range(epochs)]
    assert ep.train_graph is not None, 'the epoch never reached the captured graph'
    shares = sh_g.nnz_shares()
    err_w = max((a - b).abs().max().item() for a, b in zip(m_e.parameters(), m_g.parameters()))
    err_a = max(abs(a - b) for a, b in zip(acc_e, acc_g))
    print(f'rank {rank}: weights differ by {err_w:.3e}, accuracies by {err_a:.3e}, nnz shares {shares}', flush=True)
    ok = err_w == 0.0 and err_a == 0.0 and max(shares) < 1.1
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
