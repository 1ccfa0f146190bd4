"""Row-partitioned data-parallel GCN — one process per GPU, RCCL over xGMI.

The reference trains on one device; this is the multi-GPU path BASELINE.json asks for.  Nodes are dealt to the P ranks
by degree (the node of s-th largest degree goes to rank s mod P: ``balanced_partition``), so every rank holds the same
number of nodes AND of non-zeros of Â within a fraction of a per cent — contiguous blocks of ids gave rank 0 of 8 a
third of the non-zeros of a preferential-attachment graph, whose early ids are the hubs (2.7 x the ideal share: the
aggregations, the set-up Â_p·X and with them the whole step waited for that rank).  Rank p holds its nodes' rows of X,
y, the masks and of Â = D^-1/2 (A+I) D^-1/2 (all columns, renumbered rank-major); the weights are replicated.  Per layer:

    Z_p = H_p · Wᵀ                      local GEMM (matrix cores)
    Z   = all_gather(Z_p)               the one exchange step of the layer      [backward: reduce_scatter]
    out = Â_p · Z + b                   local HIP SpMM (csrc/dcr_gcn.hip)        [backward: Â_pᵀ · d_out]

The first layer's input is the constant feature matrix, so (as in the single-GPU model) Â_p·X is computed once — one
all-gather of X at set-up — and the layer is then a purely local GEMM: no exchange step at all for layer 1.

After backward, the weight gradients (a few tens of kilofloats) are summed with ONE all-reduce over a flat buffer:
the message is latency-bound, so one call beats one per parameter.  The loss is the masked NLL summed locally and
divided by the global number of training nodes, so the summed gradients equal the single-GPU gradients.
"""
import os

import torch
import torch.distributed as dist

from models.gcn import (GCN, RowSelection, _FirstLayerFn, first_layer_fused_ok, act_then_linear, aggregate, aggregate_rows, gcn_norm_csr, relu_dropout, spmm, spmm_pair,
                        spmm_rows)


def block_range(n, world, rank):
    per = (n + world - 1) // world
    r0 = min(rank * per, n)
    return r0, min(r0 + per, n), per


def balanced_partition(edge_index, num_nodes, world):
    """Deal the nodes to ``world`` ranks by degree: ``order`` = node ids by decreasing degree (ties: smaller id), rank p
    owns order[p::world].  Returns (owner rank, index inside the owner's block, nodes per rank incl. padding), the first
    two as int64 tensors over the node ids.  Deterministic: every rank computes the same partition from the edge list."""
    deg = torch.bincount(edge_index[0].reshape(-1), minlength=num_nodes)[:num_nodes]
    order = torch.sort(-deg, stable=True).indices
    pos = torch.empty_like(order)
    pos[order] = torch.arange(num_nodes, device=order.device)
    per = (num_nodes + world - 1) // world
    return pos % world, pos // world, per


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


def chunked_gather():
    """``DCR_DP_CHUNKED_GATHER=1``: the [Z_train | Z_eval] exchange of ``forward_pair`` as two asynchronous all-gathers, the
    second under the first half's aggregation (off by default: DESIGN §5 — it hides at most the training half's
    aggregation, and an asynchronous collective inside the captured epoch has never run on two GPUs here)."""
    return os.environ.get('DCR_DP_CHUNKED_GATHER', '0') == '1'


def _pad_rows(z, per):
    if z.shape[0] == per:
        return z
    pad = torch.zeros((per, z.shape[1]), dtype=z.dtype, device=z.device)
    pad[:z.shape[0]] = z
    return pad


class _GatherAggregatePair(torch.autograd.Function):
    """(Â_p·Z_train + b, Â_p·Z_eval + b) from the ranks' row blocks of both operands with ONE all-gather and one sweep of
    the local rows of Â (``spmm_pair``); only the training operand carries a gradient, so the backward reduce-scatter
    moves its half only.  See ``GCN.forward_pair``."""

    @staticmethod
    def forward(ctx, z_train, z_eval, bias, csr, n_total, per, group, sel_train=None, sel_eval=None):
        # sel_train / sel_eval (RowSelection of this rank's rows, both or neither): only those rows of the two outputs
        world = dist.get_world_size(group)
        f = z_train.shape[1]
        ctx.csr, ctx.group, ctx.per, ctx.rows, ctx.has_bias = csr, group, per, z_train.shape[0], bias is not None
        ctx.sel = sel_train
        if chunked_gather():
            # The exchange in two halves (round 5, ``DCR_DP_CHUNKED_GATHER=1``): the training operand's all-gather, then the
            # evaluation operand's, both asynchronous on the backend's own stream; the training half's aggregation runs
            # while the evaluation half is still on the links.  The same bits: each half is aggregated exactly as the
            # halves of the joint buffer are (``spmm_pair`` / ``dcr_spmm_csr_rows2_f32_dev``: "every block as a call of its own").
            halves, works = [], []
            for z in (z_train, z_eval):
                local = _pad_rows(z, per).contiguous()
                full = torch.empty((world * per, f), dtype=local.dtype, device=local.device)
                works.append(dist.all_gather_into_tensor(full, local, group=group, async_op=True))
                halves.append((full, local))          # (the send buffer stays alive until its wait)
            outs = []
            for (full, _), work, sel in zip(halves, works, (sel_train, sel_eval)):
                work.wait()
                if sel_train is not None:
                    outs.append(spmm_rows(csr, sel, full[:n_total], bias))
                else:
                    outs.append(spmm(csr.rowptr, csr.col, csr.val, full[:n_total], csr.n_rows, bias))
            return outs[0], outs[1]
        local = _pad_rows(torch.cat([z_train, z_eval], 1), per).contiguous()
        full = torch.empty((world * per, 2 * f), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local, group=group)
        if sel_train is not None:
            return spmm_rows(csr, sel_train, full[:n_total, :f], bias), spmm_rows(csr, sel_eval, full[:n_total, f:], bias)
        out = spmm_pair(csr.rowptr, csr.col, csr.val, full[:n_total].contiguous(), csr.n_rows, f, bias=bias)
        return out[:, :f].contiguous(), out[:, f:].contiguous()

    @staticmethod
    def backward(ctx, grad_train, grad_eval):
        csr, group, per = ctx.csr, ctx.group, ctx.per
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        grad_train = grad_train.contiguous()
        gz = None
        if ctx.needs_input_grad[0]:
            sel = ctx.sel
            if sel is None:
                gfull = spmm(csr.rowptr_t, csr.col_t, csr.val_t, grad_train, csr.n_cols)
            elif sel.n == 0:
                gfull = grad_train.new_zeros((csr.n_cols, grad_train.shape[1]))
            else:
                rp, ci, va = sel.transposed()
                gfull = spmm(rp, ci, va, grad_train, csr.n_cols)
            g = torch.zeros((world * per, gfull.shape[1]), dtype=gfull.dtype, device=gfull.device)
            g[:gfull.shape[0]] = gfull
            if dist.get_backend(group) == 'nccl':
                mine = torch.empty((per, g.shape[1]), dtype=g.dtype, device=g.device)
                dist.reduce_scatter_tensor(mine, g, group=group)
            else:  # gloo (CPU tests) has no reduce_scatter
                dist.all_reduce(g, group=group)
                mine = g[rank * per:(rank + 1) * per]
            gz = mine[:ctx.rows].contiguous()
        gb = grad_train.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gz, None, gb, None, None, None, None, None, None


class ShardedGCN(torch.nn.Module):
    """Wraps a replicated ``GCN`` and runs it on this rank's block of nodes."""

    def __init__(self, gcn: GCN, edge_index, num_nodes, group=None):
        super().__init__()
        self.gcn = gcn
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # Nodes are renumbered rank-major: node v -> owner(v) * per + index(v).  Blocks are padded to ``per`` nodes each;
        # a padding id is an isolated node with zero features, outside every mask: it changes nothing and keeps every
        # rank's block, gather and reduce-scatter the same size.
        owner, index, self.per = balanced_partition(edge_index, int(num_nodes), self.world)
        self.n_real = int(num_nodes)
        self.n = self.world * self.per
        self.r0, self.r1 = self.rank * self.per, (self.rank + 1) * self.per
        new_id = owner * self.per + index
        mine = (owner == self.rank).nonzero().squeeze(1)
        self.owned = mine[torch.sort(index[mine]).indices]   # this rank's nodes (original ids) in block order
        # rows of Â owned by this rank; normalisation uses the full graph's degrees
        self.csr = gcn_norm_csr(new_id[edge_index], None, self.n, add_self_loops=True, row_range=(self.r0, self.r1))
        self.nnz_local = int(self.csr.col.shape[0])
        for p in self.gcn.parameters():  # replicate rank 0's initialisation
            dist.broadcast(p.data, src=0, group=group)
        # One flat gradient buffer for the whole model, every parameter's .grad a view into it: backward accumulates in
        # place and the per-step all-reduce is ONE call on memory that never moves (the message is tens of kilobytes:
        # latency-bound; no torch.cat, no copy-back, and fixed addresses for a captured HIP graph).
        params = list(self.gcn.parameters())
        self._flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self._flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def shard(self, t):
        """This rank's rows of a per-node tensor (original node order), in block order, zero-padded to ``per`` rows."""
        mine = t.index_select(0, self.owned.to(t.device))
        if mine.shape[0] == self.per:
            return mine.contiguous()
        pad = torch.zeros((self.per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:mine.shape[0]] = mine
        return pad

    def nnz_shares(self):
        """Non-zeros of Â per rank as a fraction of the ideal share 1 / P (list over ranks; a collective)."""
        t = torch.zeros(self.world, dtype=torch.float64, device=self.csr.col.device)
        t[self.rank] = float(self.nnz_local)
        dist.all_reduce(t, group=self.group)
        return (t / t.sum() * self.world).tolist()

    def row_selection(self, idx):
        """``RowSelection`` of local row indices (an int64 tensor, e.g. from ``_selection``), cached per tensor."""
        if os.environ.get('DCR_GCN_ALL_ROWS', '0') == '1':
            return None
        cache = getattr(self, '_rowsel', None)
        if cache is None:
            cache = self._rowsel = {}
        key = (idx.data_ptr(), idx._version, tuple(idx.shape))
        if key not in cache:
            sel = RowSelection(self.csr, idx)
            if sel.expand is not None:   # (the local selections come from split masks; a repeated node is a caller's mistake)
                raise ValueError('row selection of the data-parallel model repeats a node')
            cache[key] = (sel, idx)
        return cache[key][0]

    def forward(self, x_local, rows=None):
        # rows (RowSelection of local rows): the log-probabilities of those rows only (models/gcn.py GCN.forward)
        h = x_local
        layers = self.gcn.layers
        for i, layer in enumerate(layers):
            last = i + 1 == len(layers)
            if i == 0 and layer.propagate_input_first and not h.requires_grad:
                h = layer.lin(self.propagated_input_local(h), layer.bias)   # (Â_p·X)·Wᵀ + b, no exchange
                if not last:
                    h = relu_dropout(h, self.gcn.act_fn, self.gcn.dropout)
                elif rows is not None:
                    h = h.index_select(0, rows.idx)
                continue
            z_local = layer.lin(h)
            z = _GatherRows.apply(z_local, self.n, self.per, self.group)
            if last and rows is not None:
                h = aggregate_rows(z, layer.bias, self.csr, rows)
            else:
                h = aggregate(z, layer.bias, self.csr)
            if not last:
                h = relu_dropout(h, self.gcn.act_fn, self.gcn.dropout)
        return torch.nn.functional.log_softmax(h, dim=1)

    def forward_pair(self, x_local, rows_train=None, rows_eval=None):
        """(training-mode, evaluation-mode) log-probabilities of this rank's nodes in one pass: one all-gather and one
        sweep of the local rows of Â per layer for both (``GCN.forward_pair``, data-parallel).  Call in training mode.
        ``rows_train`` / ``rows_eval`` (RowSelection, both or neither): those rows of the two outputs only."""
        layers = list(self.gcn.layers)
        first = layers[0]
        z_first = None
        if first.propagate_input_first and not x_local.requires_grad:
            ax = self.propagated_input_local(x_local)
            drop = self.gcn.dropout
            if (len(layers) > 1 and drop.training and 0.0 < drop.p < 1.0
                    and first_layer_fused_ok(ax, self.gcn.act_fn, first, layers[1].lin)):
                # (Â_p·X)·W1ᵀ + b1, the activation and the second layer's lin in one kernel (models/gcn.py, _FirstLayerFn)
                z_first = _FirstLayerFn.apply(ax, first.lin.weight, first.bias, layers[1].lin.weight, drop.p, True, True)
            else:
                o_tr = first.lin(ax, first.bias)
        else:
            z = _GatherRows.apply(first.lin(x_local), self.n, self.per, self.group)
            o_tr = aggregate(z, first.bias, self.csr)
        if z_first is None:
            o_ev = o_tr.detach()
        for layer in layers[1:]:
            if z_first is not None:
                (z_tr, z_ev), z_first = z_first, None
            elif o_ev.data_ptr() == o_tr.data_ptr():
                z_tr, z_ev = act_then_linear(o_tr, self.gcn.act_fn, self.gcn.dropout, layer.lin, want_train=True, want_eval=True)
            else:
                z_tr, _ = act_then_linear(o_tr, self.gcn.act_fn, self.gcn.dropout, layer.lin, want_train=True, want_eval=False)
                _, z_ev = act_then_linear(o_ev, self.gcn.act_fn, self.gcn.dropout, layer.lin, want_train=False, want_eval=True)
            if rows_train is not None and layer is layers[-1]:
                o_tr, o_ev = _GatherAggregatePair.apply(z_tr, z_ev, layer.bias, self.csr, self.n, self.per, self.group,
                                                        rows_train, rows_eval)
            else:
                o_tr, o_ev = _GatherAggregatePair.apply(z_tr, z_ev, layer.bias, self.csr, self.n, self.per, self.group)
        if rows_train is not None and len(layers) == 1:
            o_tr, o_ev = o_tr.index_select(0, rows_train.idx), o_ev.index_select(0, rows_eval.idx)
        log_softmax = torch.nn.functional.log_softmax
        return log_softmax(o_tr, dim=1), log_softmax(o_ev, dim=1)

    def _column_block(self, q):
        """(rowptr, col, val) of this rank's rows of Â restricted to the columns of rank q's block, renumbered 0..per-1."""
        csr, lo = self.csr, q * self.per
        keep = (csr.col >= lo) & (csr.col < lo + self.per)
        cs = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=keep.device)
        torch.cumsum(keep, 0, out=cs[1:])
        return cs[csr.rowptr].contiguous(), (csr.col[keep] - lo).to(csr.col.dtype).contiguous(), csr.val[keep].contiguous()

    def propagated_input_local(self, x_local):
        """Rows of Â·X owned by this rank, computed once per (x_local, graph).  Round 4: block-streamed — the peers' blocks of X
        come one at a time into ONE buffer of N / P rows (a broadcast from their owner) and each adds its column range,
        Â_p[:, block q] · X_q, to the result: the transient is two blocks (the buffer and a product), not the whole N x F
        matrix on every rank (1 GB per rank at the 1M-node bench shape, 8 GB of gathers at eight ranks before the first
        epoch).  ``DCR_DP_SETUP=gather`` restores the one-shot all-gather (same result to rounding: the row sums are formed
        block by block here)."""
        # (the keyed tensor is held in _ax_ref: its storage cannot be recycled while the entry lives)
        key = (x_local.data_ptr(), x_local._version, tuple(x_local.shape), tuple(x_local.stride()))
        if key != getattr(self, '_ax_key', None):
            with torch.no_grad():
                if os.environ.get('DCR_DP_SETUP', 'stream') == 'gather' or self.world == 1:
                    x_full = _GatherRows.apply(x_local.contiguous(), self.n, self.per, self.group)
                    self._ax = spmm(self.csr.rowptr, self.csr.col, self.csr.val, x_full.contiguous(), self.csr.n_rows)
                else:
                    mine = _pad_rows(x_local, self.per).contiguous()
                    buf = torch.empty_like(mine)
                    acc = torch.zeros((self.csr.n_rows, mine.shape[1]), dtype=mine.dtype, device=mine.device)
                    for q in range(self.world):
                        src = dist.get_global_rank(self.group, q) if self.group is not None else q
                        block = mine if q == self.rank else buf
                        dist.broadcast(block, src=src, group=self.group)
                        rp, ci, va = self._column_block(q)
                        if ci.numel():
                            acc += spmm(rp, ci, va, block, self.csr.n_rows)
                        del rp, ci, va
                    self._ax = acc
            self._ax_key = key
            self._ax_ref = x_local
        return self._ax

    def setup_transient_bytes(self, n_features, itemsize=4):
        """Peak device memory the set-up of Â_p·X holds beyond its result, per rank: the receive buffer and one block product
        (block-streamed), against the whole gathered matrix (DCR_DP_SETUP=gather)."""
        streamed = (self.per + self.csr.n_rows) * n_features * itemsize
        return {'streamed': streamed, 'gathered': self.n * n_features * itemsize}

    def _grads_are_views(self):
        base = self._flat.data_ptr()
        end = base + self._flat.numel() * self._flat.element_size()
        return all(p.grad is not None and base <= p.grad.data_ptr() < end for p in self.gcn.parameters())

    def zero_grads(self):
        """Instead of ``optimizer.zero_grad()`` (whose default drops the .grad tensors, and with them the views)."""
        if not self._grads_are_views():  # someone replaced a .grad (zero_grad(set_to_none=True), load of a checkpoint)
            off = 0
            for p in self.gcn.parameters():
                p.grad = self._flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self._flat.zero_()

    def allreduce_grads(self):
        if not self._grads_are_views():   # (never in train_step; kept correct for callers that manage .grad themselves)
            off = 0
            for p in self.gcn.parameters():
                k = p.numel()
                if p.grad is not None:
                    self._flat[off:off + k].copy_(p.grad.reshape(-1))
                else:
                    self._flat[off:off + k].zero_()
                off += k
            dist.all_reduce(self._flat, group=self.group)
            off = 0
            for p in self.gcn.parameters():
                k = p.numel()
                p.grad = self._flat[off:off + k].view_as(p)
                off += k
            return
        dist.all_reduce(self._flat, group=self.group)

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
        self.zero_grads()
        idx, y_sel, count = self._selection(train_mask_local, y_local)
        rows = self.row_selection(idx)
        logp = self(x_local, rows=rows)
        if count:
            loss = torch.nn.functional.nll_loss(logp if rows is not None else logp.index_select(0, idx), y_sel, reduction='sum')
        else:
            loss = logp.sum() * 0.0
        loss = loss / n_train_global
        loss.backward()
        self.allreduce_grads()
        optimizer.step()
        return loss.detach()

    def train_eval_step(self, optimizer, x_local, y_local, train_mask_local, val_mask_local, n_train_global):
        """One training step and, from the same pass, {correct, count} of the validation split for the weights the step
        STARTED from (the evaluation of the previous epoch: ``LaggedGraphedEpoch`` in experiment/training_loop.py)."""
        self.train()
        self.zero_grads()
        idx, y_sel, count = self._selection(train_mask_local, y_local)
        vidx, vy, vcount = self._selection(val_mask_local, y_local)
        rows_tr, rows_ev = self.row_selection(idx), self.row_selection(vidx)
        lp_train, lp_eval = self.forward_pair(x_local, rows_tr, rows_ev)
        if rows_tr is None:
            lp_train_sel, lp_eval_sel = lp_train.index_select(0, idx), (lp_eval.index_select(0, vidx) if vcount else lp_eval)
        else:
            lp_train_sel, lp_eval_sel = lp_train, lp_eval
        if count:
            loss = torch.nn.functional.nll_loss(lp_train_sel, y_sel, reduction='sum')
        else:
            loss = lp_train.sum() * 0.0
        (loss / n_train_global).backward()
        self.allreduce_grads()
        optimizer.step()
        with torch.no_grad():
            correct = (lp_eval_sel.argmax(1) == vy).sum() if vcount else lp_eval.new_zeros((), dtype=torch.long)
            stats = torch.stack([correct.double(), torch.full((), float(vcount), dtype=torch.float64, device=lp_eval.device)])
            dist.all_reduce(stats, group=self.group)
        return stats

    @torch.no_grad()
    def eval_stats(self, x_local, y_local, mask_local):
        """{correct, count} over all ranks as a 2-element float64 device tensor (no host synchronisation)."""
        self.eval()
        idx, y_sel, count = self._selection(mask_local, y_local)
        rows = self.row_selection(idx)
        logp = self(x_local, rows=rows)
        if rows is None:
            logp = logp.index_select(0, idx)
        correct = (logp.argmax(1) == y_sel).sum() if count else logp.new_zeros((), dtype=torch.long)
        stats = torch.stack([correct.double(), torch.full((), float(count), dtype=torch.float64, device=logp.device)])
        dist.all_reduce(stats, group=self.group)
        return stats

    def eval_correct(self, x_local, y_local, mask_local):
        stats = self.eval_stats(x_local, y_local, mask_local)
        return (stats[0] / stats[1].clamp(min=1)).item()


class GraphedShardedEpoch:
    """One data-parallel epoch (training step, then validation accuracy) replayed as two captured HIP graphs per rank,
    the collectives inside (RCCL kernels are stream work like any other and capture with it; gloo moves data on the host
    and cannot).  At eight ranks an epoch is well under a millisecond of kernels behind ~60 launches and three
    collectives: launch-bound when run eagerly.  The first ``WARMUP`` calls run eagerly (communicators, library handles,
    the cached Â_p·X and the Adam state come into being there); every rank must make the same calls in the same order,
    as with any collective.  Needs ``capturable=True`` on the optimiser."""

    WARMUP = 3

    def __init__(self, sharded, optimizer, x_local, y_local, train_mask_local, val_mask_local, n_train_global, lagged=True):
        # lagged: ONE graph per epoch (train_eval_step: the accuracy delivered is that of the weights the step started from)
        self.lagged = lagged
        self.val_mask = val_mask_local
        self.sh, self.opt = sharded, optimizer
        self.args = (x_local, y_local, train_mask_local, n_train_global)
        self.val = (x_local, y_local, val_mask_local)
        self.calls = 0
        self.train_graph = self.eval_graph = None
        self.stream = torch.cuda.Stream(device=x_local.device)

    @staticmethod
    def supported(sharded, optimizer, x_local):
        import os
        if os.environ.get('DCR_DP_GRAPH', '1') == '0' or not x_local.is_cuda:
            return False
        if dist.get_backend(sharded.group) != 'nccl':
            return False
        return all(g.get('capturable', False) for g in optimizer.param_groups)

    def _train(self):
        x, y, m, n_train = self.args
        if self.lagged:
            return self.sh.train_eval_step(self.opt, x, y, m, self.val_mask, n_train)
        return self.sh.train_step(self.opt, x, y, m, n_train)

    def __call__(self):
        self.calls += 1
        cur = torch.cuda.current_stream(self.args[0].device)
        if self.calls <= self.WARMUP:
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                out = self._train()
                stats = out if self.lagged else self.sh.eval_stats(*self.val)
            cur.wait_stream(self.stream)
            return (stats[0] / stats[1].clamp(min=1)).item()
        if self.train_graph is None:
            self.train_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.train_graph, stream=self.stream):
                out = self._train()
            if self.lagged:
                self.stats = out
            else:
                self.eval_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.eval_graph, stream=self.stream):
                    self.stats = self.sh.eval_stats(*self.val)
        self.train_graph.replay()
        if not self.lagged:
            self.eval_graph.replay()
        return (self.stats[0] / self.stats[1].clamp(min=1)).item()
