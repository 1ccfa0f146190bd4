"""GCN — call surface of the reference's models/gcn.py:12-44.

``GCN(dataset, hidden=[64], dropout=0.5)`` with ``.layers`` (a ModuleList of
GCNConv), ``.reg_params`` / ``.non_reg_params`` (gcn.py:22-23),
``.reset_parameters()`` and ``.forward(data) -> log_softmax [N, C]``.

``GCNConv`` restates torch_geometric 2.0.3's layer (third-party, not vendored
by the reference; call sites gcn.py:19,30,36): ``gcn_norm`` with self loops,
``lin`` = bias-free Linear with glorot init (state_dict key ``lin.weight``
[out, in]), aggregation at the target node, ``bias`` added after aggregation
(state_dict key ``bias``).  The dense contraction X·Wᵀ runs on the matrix cores
through the ROCm GEMM library; the sparse aggregation Â·(XWᵀ) + b is the
hand-written HIP kernel ``dcr_spmm_csr_f32_dev`` (csrc/dcr_gcn.hip), wrapped in
an autograd Function whose backward is the same kernel on Âᵀ.
"""
import ctypes
import os
import math
from typing import List

import torch
from torch.nn import Dropout, ModuleList, Parameter, ReLU

# 'hip': the product path (ROCm tensors only, raises otherwise).
# 'torch': plain-torch aggregation; selected explicitly by CPU-only tests (gloo) — never automatically.
_AGG_BACKEND = 'hip'


def set_aggregate_backend(name):
    global _AGG_BACKEND
    if name not in ('hip', 'torch'):
        raise ValueError(name)
    _AGG_BACKEND = name


class NormCSR:
    """Â = D^-1/2 (A + I) D^-1/2 as CSR by target row, plus the CSR of Âᵀ for the backward pass."""

    def __init__(self, rowptr, col, val, rowptr_t, col_t, val_t, n_rows, n_cols):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.rowptr_t, self.col_t, self.val_t = rowptr_t, col_t, val_t
        self.n_rows, self.n_cols = n_rows, n_cols


def _csr_from_coo(target, source, w, n_rows):
    perm = torch.argsort(target, stable=True)
    counts = torch.bincount(target, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=target.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, source[perm].to(torch.int32).contiguous(), w[perm].contiguous()


def gcn_norm_csr(edge_index, edge_weight=None, num_nodes=None, add_self_loops=True, row_range=None):
    """PyG 2.0.3 ``gcn_norm`` (flow source_to_target): weights 1 if None;
    ``add_remaining_self_loops(fill_value=1)``; deg = scatter_add(w, target);
    norm = deg^-1/2[source] * w * deg^-1/2[target] (inf -> 0).

    ``row_range=(r0, r1)`` keeps only target rows r0..r1-1 (row-partitioned Â for data parallelism);
    normalisation always uses the full graph's degrees.
    """
    row, col = edge_index[0], edge_index[1]
    n = int(num_nodes)
    dev = edge_index.device
    w = torch.ones(row.shape[0], dtype=torch.float32, device=dev) if edge_weight is None else edge_weight.float()
    if add_self_loops:
        keep = row != col
        loop_w = torch.ones(n, dtype=torch.float32, device=dev)
        if (~keep).any():
            loop_w[row[~keep]] = w[~keep]
        ar = torch.arange(n, dtype=row.dtype, device=dev)
        row = torch.cat([row[keep], ar])
        col = torch.cat([col[keep], ar])
        w = torch.cat([w[keep], loop_w])
    deg = torch.zeros(n, dtype=torch.float32, device=dev).scatter_add_(0, col, w)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    norm = dinv[row] * w * dinv[col]
    if row_range is not None:
        r0, r1 = row_range
        sel = (col >= r0) & (col < r1)
        row, col, norm = row[sel], col[sel] - r0, norm[sel]
        n_rows = r1 - r0
    else:
        n_rows = n
    rowptr, cidx, val = _csr_from_coo(col, row, norm, n_rows)          # Â: rows = targets
    rowptr_t, cidx_t, val_t = _csr_from_coo(row, col, norm, n)          # Âᵀ: rows = sources
    return NormCSR(rowptr, cidx, val, rowptr_t, cidx_t, val_t, n_rows, n)


def _spmm_hip(rowptr, col, val, B, n_rows, bias=None, relu=False):
    if not B.is_cuda:
        raise RuntimeError('GCN aggregation runs on the MI355X HIP kernel: move the model and data to a ROCm '
                           'device (there is no CPU fallback)')
    from dcr import _lib
    B = B.contiguous()
    if B.dtype != torch.float32:
        raise TypeError('dcr_spmm_csr_f32_dev is fp32')
    F = B.shape[1]
    C = torch.empty((n_rows, F), dtype=torch.float32, device=B.device)
    stream = torch.cuda.current_stream(B.device).cuda_stream
    rc = _lib.lib().dcr_spmm_csr_f32_dev(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), B.data_ptr(),
                                         C.data_ptr(), n_rows, F, F, F,
                                         bias.data_ptr() if bias is not None else None, int(relu),
                                         ctypes.c_void_p(stream))
    _lib.check(rc)
    return C


def _spmm_torch(rowptr, col, val, B, n_rows, bias=None, relu=False):
    counts = rowptr[1:] - rowptr[:-1]
    tgt = torch.repeat_interleave(torch.arange(n_rows, device=B.device), counts)
    out = torch.zeros((n_rows, B.shape[1]), dtype=B.dtype, device=B.device)
    out.index_add_(0, tgt, B[col.long()] * val[:, None])
    if bias is not None:
        out = out + bias
    return torch.relu(out) if relu else out


def spmm(rowptr, col, val, B, n_rows, bias=None, relu=False):
    fn = _spmm_hip if _AGG_BACKEND == 'hip' else _spmm_torch
    return fn(rowptr, col, val, B, n_rows, bias, relu)


def spmm_pair(rowptr, col, val, B2, n_rows, n_feat, bias=None, split=False):
    """[Â·B0 + bias | Â·B1 + bias] for B2 = [B0 | B1] (two blocks of ``n_feat`` columns) in ONE sweep of the indices
    (``dcr_spmm_csr_f32_pair_dev``): every block bit-identical to a ``spmm`` call of its own.  ``split``: the two blocks
    as two contiguous matrices (written there by the kernel: no copies afterwards)."""
    if _AGG_BACKEND != 'hip':
        a, b = (_spmm_torch(rowptr, col, val, B2[:, :n_feat], n_rows, bias), _spmm_torch(rowptr, col, val, B2[:, n_feat:], n_rows, bias))
        return (a, b) if split else torch.cat([a, b], 1)
    if not B2.is_cuda:
        raise RuntimeError('GCN aggregation runs on the MI355X HIP kernel (there is no CPU fallback)')
    from dcr import _lib
    B2 = B2.contiguous()
    stream = torch.cuda.current_stream(B2.device).cuda_stream
    if split and n_feat % 4 == 0:
        C = torch.empty((2, n_rows, n_feat), dtype=torch.float32, device=B2.device)
        _lib.check(_lib.lib().dcr_spmm_csr_f32_pair_split_dev(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), B2.data_ptr(),
                                                              C[0].data_ptr(), C[1].data_ptr(), n_rows, n_feat, 2 * n_feat, n_feat,
                                                              bias.data_ptr() if bias is not None else None, 0,
                                                              ctypes.c_void_p(stream)))
        return C[0], C[1]
    C = torch.empty((n_rows, 2 * n_feat), dtype=torch.float32, device=B2.device)
    _lib.check(_lib.lib().dcr_spmm_csr_f32_pair_dev(rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), B2.data_ptr(),
                                                    C.data_ptr(), n_rows, n_feat, 2 * n_feat, 2 * n_feat,
                                                    bias.data_ptr() if bias is not None else None, 0,
                                                    ctypes.c_void_p(stream)))
    return (C[:, :n_feat].contiguous(), C[:, n_feat:].contiguous()) if split else C


class _Aggregate(torch.autograd.Function):
    """out = Â·Z + b ; dZ = Âᵀ·dout ; db = Σ_rows dout."""

    @staticmethod
    def forward(ctx, z, bias, csr):
        ctx.csr = csr
        ctx.has_bias = bias is not None
        return spmm(csr.rowptr, csr.col, csr.val, z, csr.n_rows, bias=bias)

    @staticmethod
    def backward(ctx, grad_out):
        csr = ctx.csr
        grad_out = grad_out.contiguous()
        gz = spmm(csr.rowptr_t, csr.col_t, csr.val_t, grad_out, csr.n_cols) if ctx.needs_input_grad[0] else None
        gb = grad_out.sum(0) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gz, gb, None


def aggregate(z, bias, csr):
    return _Aggregate.apply(z, bias, csr)


class _AggregatePair(torch.autograd.Function):
    """(Â·Z_train + b, Â·Z_eval + b) in one pass over the graph; only the training operand carries a gradient."""

    @staticmethod
    def forward(ctx, z_train, z_eval, bias, csr):
        ctx.csr = csr
        ctx.has_bias = bias is not None
        f = z_train.shape[1]
        if (z_train.is_cuda and z_train.stride() == (2 * f, 1) and z_eval.stride() == (2 * f, 1)
                and z_eval.data_ptr() == z_train.data_ptr() + 4 * f and z_train.dtype == torch.float32):
            b2 = torch.as_strided(z_train, (z_train.shape[0], 2 * f), (2 * f, 1))   # the two halves of one buffer (act_then_linear)
        else:
            b2 = torch.cat([z_train, z_eval], 1)
        return spmm_pair(csr.rowptr, csr.col, csr.val, b2, csr.n_rows, f, bias=bias, split=True)

    @staticmethod
    def backward(ctx, grad_train, grad_eval):
        csr = ctx.csr
        grad_train = grad_train.contiguous()
        gz = spmm(csr.rowptr_t, csr.col_t, csr.val_t, grad_train, csr.n_cols) if ctx.needs_input_grad[0] else None
        gb = grad_train.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gz, None, gb, None


class RowSelection:
    """The rows of the last aggregation an epoch reads — the nodes of a split (experiment/training_loop.py:50-51,64-71 index
    the model's output with the split masks and use nothing else) — and, for the backward pass, Âᵀ restricted to those
    columns: the gradient of the loss is zero outside them, so dZ = Âᵀ·dOut needs the selected columns only."""

    def __init__(self, csr, rows):
        # ``expand``: None, or the positions in ``idx`` of the caller's entries when its index tensor repeats a node:
        # the kernels (and ``transposed``, whose column map holds one position per node) then work on the distinct nodes and
        # the callers gather ``out.index_select(0, expand)``, whose backward ADDS the gradients of the repeats — what
        # ``model(data)[rows].backward()`` does.  Checked once per selection (one host sync; selections are cached).
        self.expand = None
        if rows.dtype == torch.bool:
            if rows.dim() != 1 or rows.numel() != csr.n_rows:
                raise IndexError(f'row mask of shape {tuple(rows.shape)} for {csr.n_rows} nodes')
            idx = rows.nonzero().squeeze(1)
        else:
            if rows.dim() != 1 or rows.dtype not in (torch.int64, torch.int32):
                raise IndexError('rows must be a boolean node mask or a 1-d int64 / int32 index tensor')
            idx = rows.to(torch.int64)
            if idx.numel():
                lo, hi = int(idx.min()), int(idx.max())
                if lo < -csr.n_rows or hi >= csr.n_rows:
                    raise IndexError(f'row index out of range for {csr.n_rows} nodes: [{lo}, {hi}]')
                if lo < 0:
                    idx = torch.where(idx < 0, idx + csr.n_rows, idx)     # as tensor indexing reads them
                uniq, inverse = torch.unique(idx, return_inverse=True)
                if uniq.numel() != idx.numel():
                    idx, self.expand = uniq, inverse.contiguous()
        self.idx = idx.contiguous()
        self.n = int(self.idx.numel())
        self._csr = csr
        self._t = None

    def expanded(self, out):
        """``out`` (one row per distinct selected node) in the caller's index order, repeats included."""
        return out if self.expand is None else out.index_select(0, self.expand)

    def joined_with(self, other):
        """[idx | other.idx] as one tensor (the row list of ``dcr_spmm_csr_rows2_f32_dev``); built once per pair."""
        cache = self.__dict__.setdefault('_joined', {})
        hit = cache.get(id(other))
        if hit is None or hit[0] is not other:
            hit = cache[id(other)] = (other, torch.cat([self.idx, other.idx]).contiguous())
        return hit[1]

    def transposed(self):
        """(rowptr, col, val) of Âᵀ[:, idx] with the columns renumbered 0..n-1 (positions in ``idx``); built once."""
        if self._t is None:
            csr, dev = self._csr, self.idx.device
            pos = torch.full((csr.n_rows,), -1, dtype=torch.int32, device=dev)
            pos[self.idx] = torch.arange(self.n, dtype=torch.int32, device=dev)
            p = pos[csr.col_t.long()]
            keep = p >= 0
            cs = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum(keep, 0, out=cs[1:])
            self._t = (cs[csr.rowptr_t].contiguous(), p[keep].contiguous(), csr.val_t[keep].contiguous())
        return self._t


def spmm_rows(csr, sel, B, bias=None):
    """Rows ``sel.idx`` of Â·B + bias (``dcr_spmm_csr_rows_f32_dev``): [sel.n, F], every row bit-identical to that row of
    ``spmm``.  ``B`` may be a column block of a wider matrix (stride (ld, 1))."""
    F = B.shape[1]
    if _AGG_BACKEND != 'hip':
        return _spmm_torch(csr.rowptr, csr.col, csr.val, B, csr.n_rows, bias).index_select(0, sel.idx)
    if not B.is_cuda:
        raise RuntimeError('GCN aggregation runs on the MI355X HIP kernel (there is no CPU fallback)')
    if B.dtype != torch.float32:
        raise TypeError('dcr_spmm_csr_rows_f32_dev is fp32')
    if B.dim() != 2 or B.stride(1) != 1 or B.stride(0) < F:
        B = B.contiguous()
    C = torch.empty((sel.n, F), dtype=torch.float32, device=B.device)
    if sel.n == 0:
        return C
    from dcr import _lib
    stream = torch.cuda.current_stream(B.device).cuda_stream
    _lib.check(_lib.lib().dcr_spmm_csr_rows_f32_dev(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.val.data_ptr(),
                                                    sel.idx.data_ptr(), sel.n, B.data_ptr(), C.data_ptr(), F, B.stride(0), F,
                                                    bias.data_ptr() if bias is not None else None, 0, ctypes.c_void_p(stream)))
    return C


class _AggregateRows(torch.autograd.Function):
    """out = (Â·Z + b)[rows] ; dZ = Âᵀ[:, rows]·dout ; db = Σ dout."""

    @staticmethod
    def forward(ctx, z, bias, csr, sel):
        ctx.csr, ctx.sel = csr, sel
        ctx.has_bias = bias is not None
        ctx.z_shape = z.shape
        return spmm_rows(csr, sel, z, bias)

    @staticmethod
    def backward(ctx, grad_out):
        csr, sel = ctx.csr, ctx.sel
        grad_out = grad_out.contiguous()
        gz = None
        if ctx.needs_input_grad[0]:
            if sel.n == 0:
                gz = grad_out.new_zeros(ctx.z_shape)
            else:
                rp, ci, va = sel.transposed()
                gz = spmm(rp, ci, va, grad_out, csr.n_cols)
        gb = grad_out.sum(0) if ctx.has_bias and ctx.needs_input_grad[1] else None
        return gz, gb, None, None


def aggregate_rows(z, bias, csr, sel):
    return _AggregateRows.apply(z, bias, csr, sel)


class _AggregateRowsPair(torch.autograd.Function):
    """((Â·Z_train + b)[rows_train], (Â·Z_eval + b)[rows_eval]) in ONE launch when the two operands are the halves of one
    buffer (act_then_linear writes them that way): ``dcr_spmm_csr_rows2_f32_dev``, every row as ``spmm_rows`` computes it.
    Only the training operand carries a gradient (as ``_AggregateRows``)."""

    @staticmethod
    def forward(ctx, z_train, z_eval, bias, csr, sel_train, sel_eval):
        ctx.csr, ctx.sel = csr, sel_train
        ctx.has_bias = bias is not None
        ctx.z_shape = z_train.shape
        f = z_train.shape[1]
        joined = (_AGG_BACKEND == 'hip' and z_train.is_cuda and z_train.dtype == torch.float32 and f % 4 == 0
                  and z_train.stride() == (2 * f, 1) and z_eval.stride() == (2 * f, 1)
                  and z_eval.data_ptr() == z_train.data_ptr() + 4 * f and sel_train.n > 0 and sel_eval.n > 0)
        if not joined:
            out_ev = spmm_rows(csr, sel_eval, z_eval, bias)
            ctx.mark_non_differentiable(out_ev)
            return spmm_rows(csr, sel_train, z_train, bias), out_ev
        from dcr import _lib
        both = sel_train.joined_with(sel_eval)
        C = torch.empty((sel_train.n + sel_eval.n, f), dtype=torch.float32, device=z_train.device)
        stream = torch.cuda.current_stream(z_train.device).cuda_stream
        _lib.check(_lib.lib().dcr_spmm_csr_rows2_f32_dev(csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.val.data_ptr(), both.data_ptr(),
                                                         sel_train.n, sel_train.n + sel_eval.n, z_train.data_ptr(), f, C.data_ptr(), f,
                                                         2 * f, f, bias.data_ptr() if bias is not None else None, 0,
                                                         ctypes.c_void_p(stream)))
        out_tr, out_ev = C[:sel_train.n], C[sel_train.n:]
        ctx.mark_non_differentiable(out_ev)
        return out_tr, out_ev

    @staticmethod
    def backward(ctx, grad_train, grad_eval):
        csr, sel = ctx.csr, ctx.sel
        grad_train = grad_train.contiguous()
        gz = None
        if ctx.needs_input_grad[0]:
            if sel.n == 0:
                gz = grad_train.new_zeros(ctx.z_shape)
            else:
                rp, ci, va = sel.transposed()
                gz = spmm(rp, ci, va, grad_train, csr.n_cols)
        gb = grad_train.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gz, None, gb, None, None, None


_HEAD_WS = {}


def _head_workspace(device, stream):
    """Partials + tickets of the head kernels (dcr_head_*): zero before the first use, left zero by every launch — allocated
    once per (device, stream) and kept, so that the captured epoch replays on a buffer that outlives the capture."""
    from dcr import _lib
    key = (str(device), int(stream))
    ws = _HEAD_WS.get(key)
    if ws is None:
        need = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_head_workspace(ctypes.byref(need)))
        ws = _HEAD_WS[key] = torch.zeros((need.value + 7) // 8, dtype=torch.int64, device=device)
    return ws


def head_ok(z, sel_train, sel_eval, n_classes):
    """Whether the one-kernel head takes these outputs: fp32 on the GPU, at most 32 classes, row selections without repeats."""
    return (_AGG_BACKEND == 'hip' and z.is_cuda and z.dtype == torch.float32 and 1 <= n_classes <= 32
            and (sel_train is None or sel_train.expand is None) and (sel_eval is None or sel_eval.expand is None)
            and os.environ.get('DCR_FUSED_HEAD', '1') != '0')


class _AggregateRowsHead(torch.autograd.Function):
    """The last aggregation at the rows an epoch reads AND what the epoch does with them, without the log-probabilities in
    between (models/gcn.py:44 + experiment/training_loop.py:51 and :64-71):
        loss    = F.nll_loss(log_softmax((Â·Z_train + b)[rows_train]), y_train)      (float32 0-dim; None without training rows)
        correct = ((Â·Z_eval + b)[rows_eval].argmax(1) == y_eval).sum()             (int64 0-dim; None without evaluated rows)
    One aggregation launch (``dcr_spmm_csr_rows2_f32_dev`` when both operands are the halves of one buffer) and one head kernel
    (``dcr_head_fwd_f32_dev``); backward: ``dcr_head_bwd_f32_dev`` (gradient of the selected outputs and its column sums = the
    bias gradient), then dZ = Âᵀ[:, rows]·dOut.  Every selected output row is what ``spmm_rows`` computes."""

    @staticmethod
    def forward(ctx, z_train, z_eval, bias, csr, sel_train, sel_eval, y_train, y_eval):
        from dcr import _lib
        have_tr, have_ev = z_train is not None, z_eval is not None
        ref = z_train if have_tr else z_eval
        f = ref.shape[1]
        out_tr = out_ev = None
        if have_tr and have_ev:
            out_tr, out_ev = _AggregateRowsPair.forward(_NoCtx(), z_train, z_eval, bias, csr, sel_train, sel_eval)
        elif have_tr:
            out_tr = spmm_rows(csr, sel_train, z_train, bias)
        else:
            out_ev = spmm_rows(csr, sel_eval, z_eval, bias)
        stream = torch.cuda.current_stream(ref.device).cuda_stream
        ws = _head_workspace(ref.device, stream)
        loss = torch.empty((), dtype=torch.float32, device=ref.device) if have_tr else None
        correct = torch.empty((), dtype=torch.int64, device=ref.device) if have_ev else None
        m_tr = out_tr.shape[0] if have_tr else 0
        m_ev = out_ev.shape[0] if have_ev else 0
        if m_tr + m_ev > 0:
            _lib.check(_lib.lib().dcr_head_fwd_f32_dev(
                out_tr.data_ptr() if m_tr else None, out_tr.stride(0) if m_tr else f, y_train.data_ptr() if m_tr else None, m_tr,
                out_ev.data_ptr() if m_ev else None, out_ev.stride(0) if m_ev else f, y_eval.data_ptr() if m_ev else None, m_ev, f,
                loss.data_ptr() if have_tr else None, correct.data_ptr() if have_ev else None, ws.data_ptr(), ws.numel() * 8,
                ctypes.c_void_p(stream)))
        if have_tr and not m_tr:
            loss.fill_(float('nan'))          # (the mean over no rows, as F.nll_loss gives it)
        if have_ev and not m_ev:
            correct.zero_()
        ctx.csr, ctx.sel, ctx.has_bias, ctx.out_tr, ctx.y_train = csr, sel_train, bias is not None, out_tr, y_train
        ctx.z_shape = z_train.shape if have_tr else None
        if correct is not None:
            ctx.mark_non_differentiable(correct)
        ctx.set_materialize_grads(False)   # (no zero-filled "gradient" of the count: a fill launch per step)
        return loss, correct

    @staticmethod
    def backward(ctx, g_loss, _g_correct):
        from dcr import _lib
        csr, sel, out_tr = ctx.csr, ctx.sel, ctx.out_tr
        if g_loss is None or out_tr is None:
            return None, None, None, None, None, None, None, None
        m, f = out_tr.shape
        gz = gb = None
        if m == 0:
            return (out_tr.new_zeros(ctx.z_shape) if ctx.needs_input_grad[0] else None, None,
                    out_tr.new_zeros(f) if ctx.has_bias and ctx.needs_input_grad[2] else None, None, None, None, None, None)
        g = g_loss.contiguous().float()
        grad = torch.empty((m, f), dtype=torch.float32, device=out_tr.device)
        gb_buf = torch.empty(f, dtype=torch.float32, device=out_tr.device)
        stream = torch.cuda.current_stream(out_tr.device).cuda_stream
        ws = _head_workspace(out_tr.device, stream)
        _lib.check(_lib.lib().dcr_head_bwd_f32_dev(out_tr.data_ptr(), out_tr.stride(0), ctx.y_train.data_ptr(), m, f, g.data_ptr(),
                                                   grad.data_ptr(), gb_buf.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                                   ctypes.c_void_p(stream)))
        if ctx.needs_input_grad[0]:
            rp, ci, va = sel.transposed()
            gz = spmm(rp, ci, va, grad, csr.n_cols)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gb_buf
        return gz, None, gb, None, None, None, None, None


class _NoCtx:
    """Stands in for an autograd context where a Function's forward is reused as a plain function."""
    needs_input_grad = (False,) * 8

    def mark_non_differentiable(self, *a):
        pass


def atb_hip(a, b):
    """``a.t() @ b`` for tall-skinny fp32 operands [K, M], [K, N] on the hand-written MFMA kernel
    (csrc/dcr_gemm.hip, ``dcr_atb_f32_dev``)."""
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError('dcr_atb_f32_dev runs on the MI355X (there is no CPU fallback)')
    from dcr import _lib
    a, b = a.contiguous(), b.contiguous()
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape[0] != b.shape[0]:
        raise TypeError('atb_hip: fp32 [K, M] and [K, N] expected')
    K, M, N = a.shape[0], a.shape[1], b.shape[1]
    need = ctypes.c_int64()
    _lib.check(_lib.lib().dcr_atb_f32_workspace(K, M, N, ctypes.byref(need)))
    ws = torch.empty(max(need.value, 1), dtype=torch.float32, device=a.device)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    _lib.check(_lib.lib().dcr_atb_f32_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), K, M, N, M, N, N, ws.data_ptr(),
                                          need.value, ctypes.c_void_p(stream)))
    return out


class _LinearFn(torch.autograd.Function):
    """y = x·Wᵀ (+ b) through the GEMM library; dW = dyᵀ·x (a reduction over all nodes) on the hand-written MFMA
    kernel."""

    colsum_handoffs = 0   # bias gradients taken from the kernel that produced grad_out (read by the tests)

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        gx = grad_out @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            # a reduction over all nodes into a small tile: the case the GEMM library splits badly (measured 2.3x
            # and 3.9x slower at 1M nodes); for wide layers on small graphs the library is as fast
            tall = x.shape[0] >= 64 * max(x.shape[1], grad_out.shape[1])
            gw = atb_hip(grad_out, x) if tall else grad_out.t() @ x
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            ready = getattr(grad_out, '_dcr_colsum', None)   # left there by the kernel that produced this gradient
            if ready is not None and ready[:2] == (grad_out.data_ptr(), grad_out._version):
                gb = ready[2]
                _LinearFn.colsum_handoffs += 1
            else:
                gb = grad_out.sum(0)
        return gx, gw, gb


_dropout_ctr = {}  # per device: int64 call counter IN DEVICE MEMORY = offset of the Philox stream (one step per fused
#                    ReLU+dropout call).  On the device so that a captured hipGraph of the training step draws a new
#                    mask at every replay: the launch parameters stay constant, the counter moves.


def _dropout_counter(device):
    key = str(device)
    if key not in _dropout_ctr:
        _dropout_ctr[key] = torch.zeros(1, dtype=torch.int64, device=device)
    return _dropout_ctr[key]


class _ReluDropoutFn(torch.autograd.Function):
    """y = dropout(relu(x)) in one pass each way (csrc/dcr_gcn.hip), the keep mask packed to one bit per element."""

    @staticmethod
    def forward(ctx, x, p):
        from dcr import _lib
        x = x.contiguous()
        n = x.numel()
        words = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_relu_dropout_bits_words(n, ctypes.byref(words)))
        bits = torch.empty(max(words.value, 1), dtype=torch.int64, device=x.device)
        y = torch.empty_like(x)
        ctr = _dropout_counter(x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.lib().dcr_relu_dropout_fwd_f32_ctr_dev(x.data_ptr(), y.data_ptr(), bits.data_ptr(), n, float(p),
                                                               torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 0,
                                                               ctr.data_ptr(), ctypes.c_void_p(stream)))
        ctr.add_(1)
        ctx.bits, ctx.p = bits, float(p)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        from dcr import _lib
        grad_out = grad_out.contiguous()
        gin = torch.empty_like(grad_out)
        stream = torch.cuda.current_stream(grad_out.device).cuda_stream
        _lib.check(_lib.lib().dcr_relu_dropout_bwd_f32_dev(grad_out.data_ptr(), gin.data_ptr(), ctx.bits.data_ptr(),
                                                           grad_out.numel(), ctx.p, ctypes.c_void_p(stream)))
        return gin, None


def relu_dropout(x, act_fn, dropout):
    """``dropout(act_fn(x))`` (models/gcn.py:38-42).  Training on the MI355X with ReLU: the fused kernel; otherwise the
    two stock modules (evaluation, CPU tests, other activations)."""
    if (_AGG_BACKEND == 'hip' and x.is_cuda and x.dtype == torch.float32 and dropout.training and 0.0 < dropout.p < 1.0
            and isinstance(act_fn, ReLU) and x.data_ptr() % 16 == 0):
        return _ReluDropoutFn.apply(x, dropout.p)
    return dropout(act_fn(x))


class _ActLinearFn(torch.autograd.Function):
    """(dropout(relu(x))·Wᵀ, relu(x)·Wᵀ) in ONE pass over x (csrc/dcr_gcn.hip, dcr_act_linear_fwd_f32_dev): the activation
    between two layers fused into the next layer's dense contraction, for the training operand, the evaluation operand or
    both.  Only the training operand carries a gradient; its backward is one pass too (dcr_act_linear_bwd_f32_dev), the
    weight gradient the MFMA reduction kernel on the stored training activation."""

    @staticmethod
    def forward(ctx, x, weight, p, want_train, want_eval):
        from dcr import _lib
        x = x.contiguous()
        w = weight.contiguous()
        n, hidden = x.shape
        classes = w.shape[0]
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if want_train and want_eval:  # one buffer [z_train | z_eval]: what the pair aggregation reads, without a concatenation
            both = torch.empty((n, 2 * classes), dtype=x.dtype, device=x.device)
            z_tr, z_ev = both[:, :classes], both[:, classes:]
        else:
            z_tr = torch.empty((n, classes), dtype=x.dtype, device=x.device) if want_train else None
            z_ev = torch.empty((n, classes), dtype=x.dtype, device=x.device) if want_eval else None
        ldz = 2 * classes if (want_train and want_eval) else classes
        bits = ctr = None
        if want_train:
            words = ctypes.c_int64()
            _lib.check(_lib.lib().dcr_relu_dropout_bits_words(x.numel(), ctypes.byref(words)))
            bits = torch.empty(max(words.value, 1), dtype=torch.int64, device=x.device)
            ctr = _dropout_counter(x.device)
        # (no training activation is stored: the backward kernel rebuilds it from x and the keep bits)
        _lib.check(_lib.lib().dcr_act_linear_fwd_f32_dev(
            x.data_ptr(), w.data_ptr(), None, z_tr.data_ptr() if want_train else None,
            z_ev.data_ptr() if want_eval else None, ldz, bits.data_ptr() if want_train else None, n, hidden, classes, float(p),
            torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 0, ctr.data_ptr() if want_train else None, ctypes.c_void_p(stream)))
        if want_train:
            ctr.add_(1)
            ctx.save_for_backward(x, w)
            ctx.bits, ctx.p = bits, float(p)
        if want_eval:
            ctx.mark_non_differentiable(z_ev)
        return z_tr, z_ev

    @staticmethod
    def backward(ctx, g_tr, g_ev):
        from dcr import _lib
        x, w = ctx.saved_tensors
        g_tr = g_tr.contiguous()
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None, None
        # one pass over x (dcr_act_linear_bwd_fused_f32_dev): dx, its column sums — the bias gradient of the layer that
        # produced x, which _LinearFn.backward picks up instead of reading the N x H gradient once more — and dW
        stream = torch.cuda.current_stream(g_tr.device).cuda_stream
        n, hidden = x.shape
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        colsum = torch.empty(hidden, dtype=torch.float32, device=x.device)
        need = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_act_linear_bwd_fused_workspace(n, hidden, ctypes.byref(need)))
        ws = torch.empty(max(need.value, 1), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().dcr_act_linear_bwd_fused_f32_dev(g_tr.data_ptr(), w.data_ptr(), ctx.bits.data_ptr(), x.data_ptr(),
                                                               gx.data_ptr(), gw.data_ptr(), colsum.data_ptr(), ws.data_ptr(),
                                                               need.value, n, hidden, w.shape[0], ctx.p, ctypes.c_void_p(stream)))
        gx._dcr_colsum = (gx.data_ptr(), gx._version, colsum)
        return (gx if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), None, None, None


def act_then_linear(x, act_fn, dropout, lin, want_train=True, want_eval=False):
    """(lin(dropout(act_fn(x))), lin(act_fn(x))) — the part of models/gcn.py:36-42 between one layer's aggregation and the
    next one's (either may be None when not wanted).  Fused into one pass on the MI355X for ReLU, hidden width 64 / 128 and
    at most 16 classes; the stock modules otherwise (CPU tests, other shapes, other activations)."""
    w = lin.weight
    can = (_AGG_BACKEND == 'hip' and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and isinstance(act_fn, ReLU)
           and x.shape[1] in (64, 128) and w.shape[0] <= 16 and x.data_ptr() % 16 == 0)
    train_ok = can and dropout.training and 0.0 < dropout.p < 1.0
    # (each operand takes the same route whatever else is asked for in the same call: the evaluation output of the one-pass
    #  epoch equals an evaluation-mode forward bit for bit, the training output a training-mode forward)
    if want_train and want_eval and train_ok:
        return _ActLinearFn.apply(x, w, dropout.p, True, True)
    z_tr = z_ev = None
    if want_train:
        z_tr = _ActLinearFn.apply(x, w, dropout.p, True, False)[0] if train_ok else lin(relu_dropout(x, act_fn, dropout))
    if want_eval:
        with torch.no_grad():
            z_ev = _ActLinearFn.apply(x.detach(), w.detach(), 0.0, False, True)[1] if can else lin(act_fn(x.detach()))
    return z_tr, z_ev


class _DropoutAhead:
    """The dropout decisions of the NEXT training call of the one-kernel first layer, drawn on a side stream while the rest of
    the current epoch runs (dcr_dropout_words_dev; round 5).  The Philox instructions of the decisions were most of the vector
    work of k_first_layer_fwd, and on this chip vector instructions add to the matrix cores' time; drawn by a kernel of their
    own they overlap the aggregations, which leave the vector units idle (they wait on their gathers).  The words carry the
    stamp of the call they were drawn for — a call with another stamp (the counter moved: a second model, a forward without
    the expected successor) draws in line, so the result never depends on this object, only the time does.

    One per (device, rows, hidden, p); the side stream forks after the forward kernel and is joined before the next use of the
    words: at the start of the layer's backward, or by ``join_dropout_ahead()`` (the epoch drivers call it before a capture
    ends), or at the next forward."""

    def __init__(self, device, n, hidden, p):
        from dcr import _lib
        words = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_dropout_words_count(n, hidden, ctypes.byref(words)))
        self.words = torch.zeros(words.value, dtype=torch.int64, device=device)   # (stamp of no call: rows = 0)
        self.next_offset_ptr = self.words.data_ptr() + 8 * (words.value - 4)
        self.stream = torch.cuda.Stream(device)
        self.device, self.n, self.hidden, self.p = device, int(n), int(hidden), float(p)
        self.forked = False

    def draw(self):
        from dcr import _lib
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        _lib.check(_lib.lib().dcr_dropout_words_dev(self.words.data_ptr(), self.n, self.hidden, self.p,
                                                    torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 0, self.next_offset_ptr,
                                                    ctypes.c_void_p(self.stream.cuda_stream)))
        self.forked = True

    def join(self):
        if self.forked:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self.forked = False


_AHEAD = {}


def _dropout_ahead(device, n, hidden, p):
    """The _DropoutAhead of this shape, or None: not switched on (``DCR_DROPOUT_AHEAD=1``; OFF by default — measured on the
    MI355X at the 1M-node shape, the forward kernel gains 23 us of its 700 inside the epoch (82 us back to back, where the
    clock is lower) and the aggregation beside the drawing kernel loses 14: 1.585 ms per epoch against 1.589, within the noise;
    profiles/r05_dropout_ahead.txt), or it would have to be created inside a stream capture (the epoch drivers run eager
    epochs first, which create it)."""
    if os.environ.get('DCR_DROPOUT_AHEAD', '0') != '1':
        return None
    key = (str(device), int(n), int(hidden), float(p))
    ahead = _AHEAD.get(key)
    if ahead is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        if len(_AHEAD) >= 4:
            old = _AHEAD.pop(next(iter(_AHEAD)))
            old.join()
        ahead = _AHEAD[key] = _DropoutAhead(device, n, hidden, p)
    return ahead


def join_dropout_ahead():
    """Join every side stream drawing dropout decisions ahead into the current stream (before a stream capture ends)."""
    for ahead in _AHEAD.values():
        ahead.join()


class _FirstLayerFn(torch.autograd.Function):
    """The first layer's GEMM on Â·X, bias, ReLU (+ dropout) and the second layer's lin in ONE kernel on the matrix cores
    (csrc/dcr_gcn_first.hip, dcr_first_layer_fwd_f32_dev): models/gcn.py:36-42 from ``x`` of the first GCNConv to the
    second GCNConv's ``lin`` output, for the training operand, the evaluation operand or both.  The hidden activation
    never leaves the registers; the pre-activation is written once for the backward pass (training) or not at all.
    Backward: one kernel too (dcr_first_layer_bwd_f32_dev: dW1, db1, dW2 with the pre-activation's gradient in registers);
    ``DCR_FIRST_BWD_FUSED=0``: the fused MFMA pass of _ActLinearFn (dx, db1, dW2), then dW1 = dxᵀ·(Â·X)."""

    @staticmethod
    def forward(ctx, ax, w1, b1, w2, p, want_train, want_eval):
        from dcr import _lib
        w1, w2 = w1.contiguous(), w2.contiguous()
        # ax: Â·X, [n, feats] or (round 5) [n, feats rounded up to 16] with zero pad columns — what the K-chunked kernel for
        # input widths like Cora's 1,433 and Citeseer's 3,703 reads (GCNConv.propagated_input(..., pad16=True))
        if ax.stride(1) != 1 or ax.stride(0) % 4:
            ax = ax.contiguous()
        n = ax.shape[0]
        hidden, feats, classes = w1.shape[0], w1.shape[1], w2.shape[0]
        f16 = (feats + 15) // 16 * 16
        if ax.shape[1] < f16:                      # (a caller that did not pad: pad here, once per call)
            ax = torch.nn.functional.pad(ax, (0, f16 - ax.shape[1]))
        ldx = ax.stride(0)
        stream = torch.cuda.current_stream(ax.device).cuda_stream
        if want_train and want_eval:
            both = torch.empty((n, 2 * classes), dtype=ax.dtype, device=ax.device)
            z_tr, z_ev = both[:, :classes], both[:, classes:]
        else:
            z_tr = torch.empty((n, classes), dtype=ax.dtype, device=ax.device) if want_train else None
            z_ev = torch.empty((n, classes), dtype=ax.dtype, device=ax.device) if want_eval else None
        ldz = 2 * classes if (want_train and want_eval) else classes
        bits = ctr = pre = None
        if want_train:
            words = ctypes.c_int64()
            _lib.check(_lib.lib().dcr_relu_dropout_bits_words(n * hidden, ctypes.byref(words)))
            bits = torch.empty(max(words.value, 1), dtype=torch.int64, device=ax.device)
            ctr = _dropout_counter(ax.device)
            pre = torch.empty((n, hidden), dtype=ax.dtype, device=ax.device)
        ws = _first_layer_workspace(ax.device, stream, n, feats, hidden)
        ahead = _dropout_ahead(ax.device, n, hidden, p) if (want_train and p > 0.0 and n > 0) else None
        if ahead is not None:
            ahead.join()
        _lib.check(_lib.lib().dcr_first_layer_fwd_ws_f32_dev(
            ax.data_ptr(), ldx, w1.data_ptr(), None if b1 is None else b1.data_ptr(), w2.data_ptr(),
            None if pre is None else pre.data_ptr(), z_tr.data_ptr() if want_train else None,
            z_ev.data_ptr() if want_eval else None, ldz, bits.data_ptr() if want_train else None,
            None if ahead is None else ahead.words.data_ptr(), n, feats, hidden, classes,
            float(p), torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 0, ctr.data_ptr() if want_train else None,
            None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), ctypes.c_void_p(stream)))
        ctx.ahead = ahead
        if want_train:
            ctr.add_(1)
            if ahead is not None:
                ahead.draw()   # the next call's decisions, beside whatever this stream does next
            ctx.save_for_backward(ax, w2)
            ctx.pre, ctx.bits, ctx.p, ctx.has_bias, ctx.feats = pre, bits, float(p), b1 is not None, feats
        if want_eval:
            ctx.mark_non_differentiable(z_ev)
        ctx.set_materialize_grads(False)   # (the evaluation output's zero "gradient": N x classes floats filled per step, 10 us at S1M)
        return z_tr, z_ev

    @staticmethod
    def backward(ctx, g_tr, g_ev):
        from dcr import _lib
        if ctx.ahead is not None:
            ctx.ahead.join()   # (inside the same capture as the fork, when there is one)
        if g_tr is None:
            return None, None, None, None, None, None, None
        ax, w2 = ctx.saved_tensors
        pre = ctx.pre
        g_tr = g_tr.contiguous()
        stream = torch.cuda.current_stream(g_tr.device).cuda_stream
        n, hidden = pre.shape
        aligned = all(t.data_ptr() % 16 == 0 for t in (ax, g_tr, pre, ctx.bits))   # (the kernel's stage copies move 16-byte pieces)
        feats = ctx.feats
        if os.environ.get('DCR_FIRST_BWD_FUSED', '1') != '0' and aligned:
            # one kernel (dcr_first_layer_bwd_f32_dev): the gradient of the pre-activation stays in registers between the
            # contraction with W2 that forms it and the contraction with Â·X that consumes it
            gw1 = torch.empty((hidden, feats), dtype=torch.float32, device=pre.device)
            gw2 = torch.empty_like(w2)
            gb1 = torch.empty(hidden, dtype=torch.float32, device=pre.device)
            need = ctypes.c_int64()
            _lib.check(_lib.lib().dcr_first_layer_bwd_workspace(n, feats, hidden, ctypes.byref(need)))
            ws = torch.empty(max(need.value, 1), dtype=torch.float32, device=pre.device)
            _lib.check(_lib.lib().dcr_first_layer_bwd_f32_dev(g_tr.data_ptr(), w2.data_ptr(), ctx.bits.data_ptr(), pre.data_ptr(),
                                                              ax.data_ptr(), ax.stride(0), gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(),
                                                              ws.data_ptr(), need.value, n, feats, hidden, w2.shape[0], ctx.p,
                                                              ctypes.c_void_p(stream)))
            return (None, gw1 if ctx.needs_input_grad[1] else None, gb1 if (ctx.has_bias and ctx.needs_input_grad[2]) else None,
                    gw2 if ctx.needs_input_grad[3] else None, None, None, None)
        if ax.shape[1] != feats:
            ax = ax[:, :feats].contiguous()      # (the separate kernels take the unpadded matrix)
        gx = torch.empty_like(pre)
        gw2 = torch.empty_like(w2)
        colsum = torch.empty(hidden, dtype=torch.float32, device=pre.device)
        need = ctypes.c_int64()
        _lib.check(_lib.lib().dcr_act_linear_bwd_fused_workspace(n, hidden, ctypes.byref(need)))
        ws = torch.empty(max(need.value, 1), dtype=torch.float32, device=pre.device)
        _lib.check(_lib.lib().dcr_act_linear_bwd_fused_f32_dev(g_tr.data_ptr(), w2.data_ptr(), ctx.bits.data_ptr(), pre.data_ptr(),
                                                               gx.data_ptr(), gw2.data_ptr(), colsum.data_ptr(), ws.data_ptr(),
                                                               need.value, n, hidden, w2.shape[0], ctx.p, ctypes.c_void_p(stream)))
        gw1 = None
        if ctx.needs_input_grad[1]:
            tall = n >= 64 * max(ax.shape[1], hidden)   # (as _LinearFn.backward)
            gw1 = atb_hip(gx, ax) if tall else gx.t() @ ax
        gb1 = colsum if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return None, gw1, gb1, (gw2 if ctx.needs_input_grad[3] else None), None, None, None


_FIRST_WS = {}


def _first_layer_workspace(device, stream, n, feats, hidden):
    """Workspace of the K-chunked first-layer kernel (None for shapes whose W1 stays resident in LDS): partial tiles + one
    ticket per 64 rows.  The tickets must be zero before the first launch and every launch leaves them zero, so the buffer is
    allocated (zeroed) once per (device, stream, shape) and kept: no fill launch per call, and the captured epoch replays on a
    buffer that outlives the capture."""
    from dcr import _lib
    need = ctypes.c_int64()
    _lib.check(_lib.lib().dcr_first_layer_fwd_workspace(n, feats, hidden, ctypes.byref(need)))
    if need.value == 0:
        return None
    key = (str(device), int(stream), int(n), int(feats), int(hidden))
    ws = _FIRST_WS.get(key)
    if ws is None:
        if len(_FIRST_WS) >= 8:
            _FIRST_WS.pop(next(iter(_FIRST_WS)))
        ws = _FIRST_WS[key] = torch.zeros(need.value, dtype=torch.float32, device=device)
    return ws


def first_layer_fused_ok(x, act_fn, first, lin2):
    """Whether the one-kernel first layer (dcr_first_layer_fwd_ws_f32_dev) takes this model: ReLU, fp32 on the MI355X, hidden
    width 64 / 128, at most 16 classes; any input width since round 5 (W1 resident in LDS where it fits, streamed through it
    in K chunks otherwise: Cora 1,433, Citeseer 3,703).  ``DCR_FIRST_FUSED=0`` switches it off."""
    if not (_AGG_BACKEND == 'hip' and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and isinstance(act_fn, ReLU)):
        return False
    if os.environ.get('DCR_FIRST_FUSED', '1') == '0':
        return False
    from dcr import _lib
    return bool(_lib.lib().dcr_first_layer_fits(int(first.lin.weight.shape[1]), int(first.lin.weight.shape[0]), int(lin2.weight.shape[0])))


class SparseInput:
    """The node features as a sparse operand (round 5).  Planetoid features are row-normalised bags of words, ~0.9 % dense
    (experiment/data_loader.py reads them; bench.py generates that density): X·W1ᵀ over the non-zeros is ~100 x fewer products
    than the dense contraction of the propagated Â·X (which is ~4-5 % dense), so for such inputs the first layer keeps PyG's
    order of operations — lin first, then the aggregation — with X in CSR and, for the weight gradient, Xᵀ in CSR:
        forward   pre = Â·(X·W1ᵀ) + b1            two aggregations (dcr_spmm_csr_f32_dev), the first over X's non-zeros
        backward  dW1ᵀ = Xᵀ·(Âᵀ·dpre)            two more
    Built once per (x, graph) by ``GCNConv.sparse_input``."""

    def __init__(self, x):
        n, f = x.shape
        idx = x.nonzero()                                   # row-major order
        rows, cols = idx[:, 0], idx[:, 1]
        self.n, self.f, self.nnz = n, f, int(rows.numel())
        self.val = x[rows, cols].contiguous()
        self.rowptr = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(torch.bincount(rows, minlength=n), 0, out=self.rowptr[1:])
        self.col = cols.to(torch.int32).contiguous()
        order = torch.argsort(cols * n + rows)              # Xᵀ: by column, rows ascending inside one
        self.rowptr_t = torch.zeros(f + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(torch.bincount(cols, minlength=f), 0, out=self.rowptr_t[1:])
        self.col_t = rows[order].to(torch.int32).contiguous()
        self.val_t = self.val[order].contiguous()


class _SparseFirstFn(torch.autograd.Function):
    """pre = Â·(X·W1ᵀ) + b1 with X sparse (``SparseInput``); see there."""

    @staticmethod
    def forward(ctx, w1, b1, xs, csr):
        w1t = w1.t().contiguous()                           # [F, H]: the rows the non-zeros of X gather
        h = spmm(xs.rowptr, xs.col, xs.val, w1t, xs.n)
        ctx.xs, ctx.csr, ctx.has_bias = xs, csr, b1 is not None
        return spmm(csr.rowptr, csr.col, csr.val, h, csr.n_rows, bias=b1)

    @staticmethod
    def backward(ctx, g):
        xs, csr = ctx.xs, ctx.csr
        g = g.contiguous()
        gw1 = gb1 = None
        if ctx.needs_input_grad[0]:
            dh = spmm(csr.rowptr_t, csr.col_t, csr.val_t, g, csr.n_cols)
            gw1 = spmm(xs.rowptr_t, xs.col_t, xs.val_t, dh, xs.f).t()    # [H, F] as a view of the [F, H] product
        if ctx.has_bias and ctx.needs_input_grad[1]:
            ready = getattr(g, '_dcr_colsum', None)         # left there by the kernel that produced this gradient
            if ready is not None and ready[:2] == (g.data_ptr(), g._version):
                gb1 = ready[2]
                _LinearFn.colsum_handoffs += 1
            else:
                gb1 = g.sum(0)
        return gw1, gb1, None, None


def sparse_input_density():
    """Largest share of non-zeros for which the first layer takes the sparse-input route (``DCR_SPARSE_X``, default 0.1;
    0 switches it off)."""
    try:
        return float(os.environ.get('DCR_SPARSE_X', '0.1'))
    except ValueError:
        return 0.1


class _Linear(torch.nn.Module):
    """torch_geometric.nn.dense.linear.Linear(in, out, bias=False, weight_initializer='glorot')."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = Parameter(torch.empty(out_channels, in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))  # glorot
        with torch.no_grad():
            self.weight.uniform_(-a, a)

    def forward(self, x, bias=None):
        if _AGG_BACKEND == 'hip' and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
            return _LinearFn.apply(x, self.weight, bias)
        return torch.nn.functional.linear(x, self.weight, bias)


class GCNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, add_self_loops=True, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.add_self_loops = add_self_loops
        self.lin = _Linear(in_channels, out_channels)
        if bias:
            self.bias = Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        self._cache_key = None
        self._cache_csr = None
        self._cache_ref = None   # the tensors the cache is keyed on (held: their storage cannot be recycled meanwhile)
        # Â·(X·Wᵀ) = (Â·X)·Wᵀ: when the layer's input is a constant of the run (the node features, for the first layer
        # of a GCN) Â·X is computed ONCE and the layer becomes a plain GEMM: no aggregation in the forward pass, none in
        # the backward pass, and in the data-parallel model no exchange step for this layer.  GCN switches it on for
        # layers[0]; a stand-alone GCNConv keeps PyG's order of operations.
        self.propagate_input_first = False
        self._ax_key = None
        self._ax = None
        self._ax_ref = None
        self._xs_key = self._xs = self._xs_ref = None
        self._rowsel = {}

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()
        self.invalidate()

    def propagated_input(self, x, csr, pad16=False):
        """Â·x, cached while x (same storage, same version) and the graph stay the same.  The cache holds a reference
        to the tensor it is keyed on: its storage cannot be freed and handed to another tensor while the entry lives,
        so an equal (address, version, shape) key always means the same values.  ``pad16``: the width rounded up to a
        multiple of 16 with zero columns (what the one-kernel first layer reads: 16-column MFMA steps, 16-byte pieces)."""
        key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), str(x.device), self._cache_key, bool(pad16))
        if key != self._ax_key:
            with torch.no_grad():
                xc = x.contiguous()
                pad = (-xc.shape[1]) % 16 if pad16 else 0
                if pad:
                    xc = torch.nn.functional.pad(xc, (0, pad))
                self._ax = spmm(csr.rowptr, csr.col, csr.val, xc, csr.n_rows)
            self._ax_key = key
            self._ax_ref = x
        return self._ax

    def sparse_input(self, x):
        """``SparseInput`` of x when the first layer should take the sparse route (few non-zeros, fp32 on the GPU, HIP backend),
        else None; decided and built once per tensor (same storage, same version: the tensor is held), one host sync."""
        thr = sparse_input_density()
        if not (thr > 0 and _AGG_BACKEND == 'hip' and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] >= 64
                and not x.requires_grad):
            return None
        key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()), str(x.device), thr)
        if key != self._xs_key:
            dense = int(torch.count_nonzero(x))
            self._xs = SparseInput(x) if dense <= thr * x.numel() else None
            self._xs_key, self._xs_ref = key, x
        return self._xs

    def invalidate(self):
        """Drop the cached Â and Â·x (they are rebuilt at the next forward)."""
        self._cache_key = self._cache_csr = self._cache_ref = None
        self._ax_key = self._ax = self._ax_ref = None
        self._xs_key = self._xs = self._xs_ref = None
        self._rowsel = {}

    def row_selection(self, rows, csr):
        """``RowSelection`` of a split (boolean mask or index tensor), cached while the tensor (same storage, same
        version: held here) and the graph stay the same."""
        key = (rows.data_ptr(), rows._version, tuple(rows.shape), rows.dtype, str(rows.device), self._cache_key)
        hit = self._rowsel.get(key)
        if hit is None:
            if len(self._rowsel) >= 6:
                self._rowsel.pop(next(iter(self._rowsel)))
            hit = self._rowsel[key] = (RowSelection(csr, rows), rows)
        return hit[0]

    def norm_csr(self, edge_index, edge_weight, num_nodes):
        # (PyG 2.0.3 with cached=False renormalises at every call; same result, the graph is a constant of a training
        #  run.  The keyed tensors are held, see propagated_input.)
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), tuple(edge_index.stride()),
               int(num_nodes),
               None if edge_weight is None else (edge_weight.data_ptr(), edge_weight._version, tuple(edge_weight.shape)),
               str(edge_index.device))
        if key != self._cache_key:
            self._cache_csr = gcn_norm_csr(edge_index, edge_weight, num_nodes, self.add_self_loops)
            self._cache_key = key
            self._cache_ref = (edge_index, edge_weight)
        return self._cache_csr

    def forward(self, x, edge_index, edge_weight=None):
        csr = self.norm_csr(edge_index, edge_weight, x.shape[0])
        if self.propagate_input_first and not x.requires_grad and x.dtype == torch.float32:
            xs = self.sparse_input(x)
            if xs is not None:                                           # Â·(X·Wᵀ) + b over X's non-zeros (SparseInput)
                return _SparseFirstFn.apply(self.lin.weight, self.bias, xs, csr)
            return self.lin(self.propagated_input(x, csr), self.bias)   # (Â·X)·Wᵀ + b: one GEMM
        z = self.lin(x)                       # dense contraction on MFMA via the GEMM library
        return aggregate(z, self.bias, csr)   # sparse aggregation + bias: HIP kernel


class GCN(torch.nn.Module):
    """The reference's model (models/gcn.py:12-44) on the kernels of this package: GCNConv layers of widths
    [features] + hidden + [classes], ReLU and dropout between them, log-softmax at the end.  Attribute names, the
    two optimiser groups (weight decay on the first layer only, save_models.py:78-82) and the ``state_dict`` keys
    (``layers.{i}.bias``, ``layers.{i}.lin.weight``) are the reference's."""

    def __init__(self, dataset, hidden: List[int] = [64], dropout: float = 0.5):
        super().__init__()
        widths = [dataset.data.x.shape[1], *hidden, dataset.num_classes]
        self.layers = ModuleList([GCNConv(w_in, w_out) for w_in, w_out in zip(widths, widths[1:])])
        first = self.layers[0]
        first.propagate_input_first = True   # its input is data.x, constant over the run
        self.reg_params = list(first.parameters())
        self.non_reg_params = [p for conv in list(self.layers)[1:] for p in conv.parameters()]
        self.dropout = Dropout(p=dropout)
        self.act_fn = ReLU()

    def reset_parameters(self):
        for conv in self.layers:
            conv.reset_parameters()

    def _fused_first(self, data, want_train, want_eval):
        """(z_train, z_eval) of the SECOND layer's lin straight from Â·X (one kernel, _FirstLayerFn), or None when the
        shapes, the mode or the backend ask for the separate kernels: each operand then takes the route it would take alone,
        so training, evaluation and the one-pass epoch see the same numbers whichever of them asks."""
        first, second = self.layers[0], self.layers[1]
        x = data.x
        if not (first.propagate_input_first and not x.requires_grad and x.dtype == torch.float32 and x.is_cuda):
            return None
        p = self.dropout.p
        if want_train and not (self.dropout.training and 0.0 < p < 1.0):
            return None
        grads = torch.is_grad_enabled() and any(q.requires_grad for q in list(first.parameters()) + [second.lin.weight])
        if want_eval and not want_train and grads:
            return None                                   # evaluation mode WITH a gradient: the stock modules
        if not first_layer_fused_ok(x, self.act_fn, first, second.lin):
            return None
        if first.sparse_input(x) is not None:
            return None                                   # few non-zeros: Â·(X·W1ᵀ) over them (GCNConv.forward), then act_then_linear
        csr = first.norm_csr(data.edge_index, data.edge_attr, x.shape[0])
        ax = first.propagated_input(x, csr, pad16=True)
        if want_train:
            return _FirstLayerFn.apply(ax, first.lin.weight, first.bias, second.lin.weight, p, True, want_eval)
        with torch.no_grad():
            return _FirstLayerFn.apply(ax, first.lin.weight.detach(), None if first.bias is None else first.bias.detach(),
                                       second.lin.weight.detach(), 0.0, False, True)

    supports_rows = True   # forward(data, rows=...) / forward_pair(data, rows_train=..., rows_eval=...)

    def forward(self, data, rows=None):
        # models/gcn.py:32-44.  The activation after a layer is computed together with the next layer's lin
        # (act_then_linear: one pass over the hidden activation where the fused kernel applies).
        # rows (a boolean node mask or an index tensor): return the log-probabilities of those nodes only, [len, C] in
        # index order — what ``model(data)[rows]`` holds, value for value; the last aggregation is evaluated at those
        # rows and nowhere else.
        layers = list(self.layers)
        z_first = self._fused_first(data, self.training, not self.training) if len(layers) > 1 else None
        if z_first is None:
            h = layers[0](data.x, data.edge_index, edge_weight=data.edge_attr)
            if rows is not None and len(layers) == 1:
                h = h[rows] if rows.dtype == torch.bool else h.index_select(0, rows)
        for conv in layers[1:]:
            if z_first is not None:          # first layer, activation and this layer's lin came out of one kernel
                z, n_nodes = z_first[0 if self.training else 1], z_first[0 if self.training else 1].shape[0]
                z_first = None
            else:
                n_nodes = h.shape[0]
                if self.training:
                    z, _ = act_then_linear(h, self.act_fn, self.dropout, conv.lin, want_train=True, want_eval=False)
                elif torch.is_grad_enabled() and h.requires_grad:
                    z = conv.lin(self.dropout(self.act_fn(h)))   # evaluation mode WITH a gradient (not on the training path)
                else:
                    _, z = act_then_linear(h, self.act_fn, self.dropout, conv.lin, want_train=False, want_eval=True)
            csr = conv.norm_csr(data.edge_index, data.edge_attr, n_nodes)
            if rows is not None and conv is layers[-1]:
                sel = conv.row_selection(rows, csr)
                h = sel.expanded(aggregate_rows(z, conv.bias, csr, sel))
            else:
                h = aggregate(z, conv.bias, csr)
        return torch.nn.functional.log_softmax(h, dim=1)

    def forward_pair(self, data, rows_train=None, rows_eval=None):
        """(training-mode log-probabilities with their autograd graph, evaluation-mode log-probabilities) of the SAME
        weights in one pass: what ``model.train(); model(data)`` and ``model.eval(); model(data)`` return, value for
        value.  The two differ only in the dropout between the layers, so the first layer's output is computed once and
        every later aggregation serves both operands in one sweep of the graph (``spmm_pair``).  The validation forward
        of one epoch and the training forward of the next see the same weights (experiment/training_loop.py:25-26:
        train, then evaluate, then train again), which is what ``LaggedGraphedEpoch`` builds on.  Call in training mode.
        ``rows_train`` / ``rows_eval`` (both or neither): only those nodes' rows of the two outputs, as ``forward(data, rows)``."""
        if (rows_train is None) != (rows_eval is None):
            raise ValueError('rows_train and rows_eval go together')
        last = len(self.layers) - 1
        first = self.layers[0]
        z_first = self._fused_first(data, True, True) if last > 0 else None
        if z_first is None:
            o_tr = first(data.x, data.edge_index, edge_weight=data.edge_attr)
            o_ev = o_tr.detach()
        for depth, conv in enumerate(list(self.layers)[1:], start=1):
            if z_first is not None:                      # first layer, activation and this layer's lin out of one kernel
                (z_tr, z_ev), z_first = z_first, None
            elif o_ev.data_ptr() == o_tr.data_ptr():     # the same pre-activation (first hidden layer): one pass for both
                z_tr, z_ev = act_then_linear(o_tr, self.act_fn, self.dropout, conv.lin, want_train=True, want_eval=True)
            else:                                        # (dropout is the identity in evaluation mode)
                z_tr, _ = act_then_linear(o_tr, self.act_fn, self.dropout, conv.lin, want_train=True, want_eval=False)
                _, z_ev = act_then_linear(o_ev, self.act_fn, self.dropout, conv.lin, want_train=False, want_eval=True)
            csr = conv.norm_csr(data.edge_index, data.edge_attr, z_tr.shape[0])
            if rows_train is not None and depth == last:
                sel_tr, sel_ev = conv.row_selection(rows_train, csr), conv.row_selection(rows_eval, csr)
                o_tr, o_ev = _AggregateRowsPair.apply(z_tr, z_ev, conv.bias, csr, sel_tr, sel_ev)
                o_tr, o_ev = sel_tr.expanded(o_tr), sel_ev.expanded(o_ev)
            else:
                o_tr, o_ev = _AggregatePair.apply(z_tr, z_ev, conv.bias, csr)
        if rows_train is not None and last == 0:
            pick = lambda t, r: t[r] if r.dtype == torch.bool else t.index_select(0, r)
            o_tr, o_ev = pick(o_tr, rows_train), pick(o_ev, rows_eval)
        log_softmax = torch.nn.functional.log_softmax
        return log_softmax(o_tr, dim=1), log_softmax(o_ev, dim=1)


def _forward_head(self, data, rows_train=None, y_train=None, rows_eval=None, y_eval=None):
    """(loss, correct) of one epoch's two readings of the model in ONE pass, without the log-probabilities in between:
    ``loss = F.nll_loss(model(data)[rows_train], y_train)`` in training mode (dropout on, autograd graph attached) and
    ``correct = (model(data)[rows_eval].argmax(1) == y_eval).sum()`` in evaluation mode, of the SAME weights (see
    ``forward_pair``).  Either half may be left out (rows_* = None): the training step alone, or the evaluation alone.
    Returns None when a shape, a mode or the backend asks for the separate kernels (the caller then takes the log-probabilities
    from ``forward`` / ``forward_pair``): two or more layers, row index tensors without repeats, at most 32 classes."""
    want_tr, want_ev = rows_train is not None, rows_eval is not None
    last = len(self.layers) - 1
    if last < 1 or not (want_tr or want_ev) or (want_tr and not self.training):
        return None
    conv = self.layers[last]
    first = self.layers[0]
    z_first = self._fused_first(data, want_tr, want_ev)
    if z_first is None:
        if want_tr:
            o_tr = first(data.x, data.edge_index, edge_weight=data.edge_attr)
            o_ev = o_tr.detach()
        else:
            with torch.no_grad():
                o_ev = first(data.x, data.edge_index, edge_weight=data.edge_attr)
            o_tr = None
    z_tr = z_ev = None
    for depth, layer in enumerate(list(self.layers)[1:], start=1):
        if z_first is not None:
            (z_tr, z_ev), z_first = z_first, None
        elif want_tr and want_ev and o_ev.data_ptr() == o_tr.data_ptr():
            z_tr, z_ev = act_then_linear(o_tr, self.act_fn, self.dropout, layer.lin, want_train=True, want_eval=True)
        else:
            z_tr = act_then_linear(o_tr, self.act_fn, self.dropout, layer.lin, want_train=True, want_eval=False)[0] if want_tr else None
            if want_ev:
                with torch.no_grad():
                    z_ev = act_then_linear(o_ev, self.act_fn, self.dropout, layer.lin, want_train=False, want_eval=True)[1]
        ref = z_tr if want_tr else z_ev
        csr = layer.norm_csr(data.edge_index, data.edge_attr, ref.shape[0])
        if depth == last:
            sel_tr = layer.row_selection(rows_train, csr) if want_tr else None
            sel_ev = layer.row_selection(rows_eval, csr) if want_ev else None
            if not head_ok(ref, sel_tr, sel_ev, ref.shape[1]):
                return None
            return _AggregateRowsHead.apply(z_tr, z_ev, layer.bias, csr, sel_tr, sel_ev, y_train, y_eval)
        if want_tr and want_ev:
            o_tr, o_ev = _AggregatePair.apply(z_tr, z_ev, layer.bias, csr)
        elif want_tr:
            o_tr = aggregate(z_tr, layer.bias, csr)
        else:
            with torch.no_grad():
                o_ev = aggregate(z_ev, layer.bias, csr)
    return None


GCN.forward_head = _forward_head


def dense_reference_logits(model, x, edge_index, num_nodes):
    """Dense fp64 restatement Â = D^-1/2 (A+I) D^-1/2, log_softmax(Â·relu(Â·X·W1ᵀ+b1)·W2ᵀ+b2) in eval mode —
    the build's own pin for GCNConv."""
    n = num_nodes
    A = torch.zeros((n, n), dtype=torch.float64, device=x.device)
    A[edge_index[1], edge_index[0]] = 1.0
    A = A + torch.eye(n, dtype=torch.float64, device=x.device)
    dinv = A.sum(1).pow(-0.5)
    Ah = dinv[:, None] * A * dinv[None, :]
    h = x.double()
    for i, layer in enumerate(model.layers):
        h = Ah @ (h @ layer.lin.weight.double().t()) + layer.bias.double()
        if i < len(model.layers) - 1:
            h = torch.relu(h)
    return torch.log_softmax(h, dim=1)


def smoke_gcn(edge_index, num_nodes, device='cuda:0'):
    from dcr.data import Data, Dataset
    torch.manual_seed(0)
    x = torch.randn(num_nodes, 32)
    y = torch.randint(0, 5, (num_nodes,))
    data = Data(x=x, edge_index=edge_index, y=y, num_nodes=num_nodes).to(device)
    model = GCN(Dataset(data, 5), hidden=[16], dropout=0.5).to(device)
    model.eval()
    got = model(data)
    want = dense_reference_logits(model, data.x, data.edge_index, num_nodes)
    err = (got.double() - want).abs().max().item()
    assert err < 1e-5, f'GCN logits differ from the dense reference by {err}'
    model.train()
    loss = torch.nn.functional.nll_loss(model(data), data.y)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    return err
