"""Row-partitioned data-parallel GCN — one process per GPU, RCCL over xGMI.

The reference trains on one device; this is the multi-GPU path BASELINE.json asks for.  Nodes are split into P
contiguous blocks.  Rank p holds its block of X, y, the masks and the rows of Â = D^-1/2 (A+I) D^-1/2 that belong to
its nodes (all columns); the weights are replicated.  Per layer:

    Z_p = H_p · Wᵀ                      local GEMM (matrix cores)
    Z   = all_gather(Z_p)               the one exchange step of the layer      [backward: reduce_scatter]
    out = Â_p · Z + b                   local HIP SpMM (csrc/dcr_gcn.hip)        [backward: Â_pᵀ · d_out]

The first layer's input is the constant feature matrix, so (as in the single-GPU model) Â_p·X is computed once — one
all-gather of X at set-up — and the layer is then a purely local GEMM: no exchange step at all for layer 1.

After backward, the weight gradients (a few tens of kilofloats) are summed with ONE all-reduce over a flat buffer:
the message is latency-bound, so one call beats one per parameter.  The loss is the masked NLL summed locally and
divided by the global number of training nodes, so the summed gradients equal the single-GPU gradients.
"""
import torch
import torch.distributed as dist

from models.gcn import GCN, aggregate, gcn_norm_csr, relu_dropout


def block_range(n, world, rank):
    per = (n + world - 1) // world
    r0 = min(rank * per, n)
    return r0, min(r0 + per, n), per


class _GatherRows(torch.autograd.Function):
    """all_gather of equally padded row blocks; backward is the matching reduce_scatter (sum)."""

    @staticmethod
    def forward(ctx, z_local, n_total, per, group):
        world = dist.get_world_size(group)
        ctx.group, ctx.per, ctx.rows = group, per, z_local.shape[0]
        pad = z_local
        if z_local.shape[0] < per:
            pad = torch.zeros((per, z_local.shape[1]), dtype=z_local.dtype, device=z_local.device)
            pad[:z_local.shape[0]] = z_local
        out = torch.empty((world * per, z_local.shape[1]), dtype=z_local.dtype, device=z_local.device)
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
        return out[:n_total]

    @staticmethod
    def backward(ctx, grad_full):
        group, per = ctx.group, ctx.per
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        g = torch.zeros((world * per, grad_full.shape[1]), dtype=grad_full.dtype, device=grad_full.device)
        g[:grad_full.shape[0]] = grad_full
        if dist.get_backend(group) == 'nccl':
            mine = torch.empty((per, g.shape[1]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(mine, g, group=group)
        else:  # gloo (CPU tests) has no reduce_scatter
            dist.all_reduce(g, group=group)
            mine = g[rank * per:(rank + 1) * per]
        return mine[:ctx.rows].contiguous(), None, None, None


class ShardedGCN(torch.nn.Module):
    """Wraps a replicated ``GCN`` and runs it on this rank's block of nodes."""

    def __init__(self, gcn: GCN, edge_index, num_nodes, group=None):
        super().__init__()
        self.gcn = gcn
        self.group = group
        self.n = int(num_nodes)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.r0, self.r1, self.per = block_range(self.n, self.world, self.rank)
        # rows of Â owned by this rank; normalisation uses the full graph's degrees
        self.csr = gcn_norm_csr(edge_index, None, self.n, add_self_loops=True, row_range=(self.r0, self.r1))
        for p in self.gcn.parameters():  # replicate rank 0's initialisation
            dist.broadcast(p.data, src=0, group=group)
        self._flat = None

    def shard(self, t):
        return t[self.r0:self.r1]

    def forward(self, x_local):
        h = x_local
        layers = self.gcn.layers
        for i, layer in enumerate(layers):
            if i == 0 and layer.propagate_input_first and not h.requires_grad:
                h = layer.lin(self.propagated_input_local(h), layer.bias)   # (Â_p·X)·Wᵀ + b, no exchange
                if i + 1 < len(layers):
                    h = relu_dropout(h, self.gcn.act_fn, self.gcn.dropout)
                continue
            z_local = layer.lin(h)
            z = _GatherRows.apply(z_local, self.n, self.per, self.group)
            h = aggregate(z, layer.bias, self.csr)
            if i + 1 < len(layers):
                h = relu_dropout(h, self.gcn.act_fn, self.gcn.dropout)
        return torch.nn.functional.log_softmax(h, dim=1)

    def propagated_input_local(self, x_local):
        """Rows of Â·X owned by this rank, computed once per (x_local, graph): all-gather X, one local SpMM."""
        # (the keyed tensor is held in _ax_ref: its storage cannot be recycled while the entry lives)
        key = (x_local.data_ptr(), x_local._version, tuple(x_local.shape), tuple(x_local.stride()))
        if key != getattr(self, '_ax_key', None):
            with torch.no_grad():
                x_full = _GatherRows.apply(x_local.contiguous(), self.n, self.per, self.group)
                from models.gcn import spmm
                self._ax = spmm(self.csr.rowptr, self.csr.col, self.csr.val, x_full.contiguous(), self.csr.n_rows)
            self._ax_key = key
            self._ax_ref = x_local
        return self._ax

    def allreduce_grads(self):
        params = [p for p in self.gcn.parameters() if p.grad is not None]
        if not params:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, group=self.group)
        off = 0
        for p in params:
            k = p.grad.numel()
            p.grad.copy_(flat[off:off + k].view_as(p.grad))
            off += k

    def _selection(self, mask_local, y_local):
        """Index tensor and labels of a mask, computed once per mask: boolean-mask indexing costs a host sync (and a
        data-dependent shape) at every use, and an epoch of this model is launch-bound on more than a few GPUs."""
        key = (mask_local.data_ptr(), mask_local._version, y_local.data_ptr(), y_local._version)
        cache = getattr(self, '_sel_cache', None)
        if cache is None:
            cache = self._sel_cache = {}
        if key not in cache:
            idx = mask_local.nonzero().squeeze(1)
            cache[key] = (idx, y_local.index_select(0, idx), int(idx.numel()), (mask_local, y_local))  # refs held
        return cache[key][:3]

    def train_step(self, optimizer, x_local, y_local, train_mask_local, n_train_global):
        self.train()
        optimizer.zero_grad()
        logp = self(x_local)
        idx, y_sel, count = self._selection(train_mask_local, y_local)
        if count:
            loss = torch.nn.functional.nll_loss(logp.index_select(0, idx), y_sel, reduction='sum')
        else:
            loss = logp.sum() * 0.0
        loss = loss / n_train_global
        loss.backward()
        self.allreduce_grads()
        optimizer.step()
        return loss.detach()

    @torch.no_grad()
    def eval_correct(self, x_local, y_local, mask_local):
        self.eval()
        logp = self(x_local)
        idx, y_sel, count = self._selection(mask_local, y_local)
        correct = (logp.index_select(0, idx).argmax(1) == y_sel).sum() if count else logp.new_zeros((), dtype=torch.long)
        stats = torch.stack([correct.double(), torch.tensor(float(count), dtype=torch.float64, device=logp.device)])
        dist.all_reduce(stats, group=self.group)
        return (stats[0] / stats[1].clamp(min=1)).item()
