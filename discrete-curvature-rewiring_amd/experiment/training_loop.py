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


class _PickedMean(torch.autograd.Function):
    """``-log_probs.gather(1, y[:, None]).mean()`` and its backward as one kernel each (csrc/dcr_gcn.hip): the gradient is
    ``-g / m`` at the picked entries and zero elsewhere, bit for bit what ``F.nll_loss`` (training_loop.py:51) gives for class ids
    in [0, C) (the callers check that once: ``GraphedEpoch._labels_fit``).  The loss VALUE is summed in float64 and rounded
    once: equal to the stock op's to the last ulp, not necessarily bit for bit."""

    @staticmethod
    def forward(ctx, log_probs, y):
        import ctypes
        from dcr import _lib
        lp = log_probs.contiguous()
        out = torch.empty((), dtype=torch.float32, device=lp.device)
        stream = torch.cuda.current_stream(lp.device).cuda_stream
        from models.gcn import _head_workspace
        _lib.check(_lib.lib().dcr_nll_picked_mean_fwd_f32_dev(lp.data_ptr(), lp.shape[1], y.data_ptr(), lp.shape[0], lp.shape[1],
                                                              out.data_ptr(), _head_workspace(lp.device, stream).data_ptr(),
                                                              ctypes.c_void_p(stream)))
        ctx.y, ctx.shape = y, tuple(lp.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from dcr import _lib
        g = g.contiguous().float()
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        stream = torch.cuda.current_stream(g.device).cuda_stream
        _lib.check(_lib.lib().dcr_nll_picked_mean_bwd_f32_dev(ctx.y.data_ptr(), ctx.shape[0], ctx.shape[1], g.data_ptr(), grad.data_ptr(),
                                                              ctypes.c_void_p(stream)))
        return grad, None


def _count_correct(log_probs, y):
    """``log_probs.max(1)[1].eq(y).sum()`` (training_loop.py:64-71) as one kernel: int64 0-dim tensor."""
    import ctypes
    from dcr import _lib
    lp = log_probs.contiguous()
    out = torch.empty((), dtype=torch.int64, device=lp.device)
    stream = torch.cuda.current_stream(lp.device).cuda_stream
    _lib.check(_lib.lib().dcr_count_argmax_equal_f32_dev(lp.data_ptr(), lp.shape[1] if lp.dim() == 2 else 1, y.data_ptr(), lp.shape[0],
                                                         lp.shape[1], out.data_ptr(), ctypes.c_void_p(stream)))
    return out


def _fused_ok(t):
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] > 0


def _takes_rows(model):
    """Models of this package compute ``model(data)[mask]`` as ``model(data, rows=mask)``: the last aggregation evaluated at
    the rows the caller reads (models/gcn.py).  ``DCR_GCN_ALL_ROWS=1``: always the full output, indexed afterwards."""
    return getattr(model, 'supports_rows', False) and os.environ.get('DCR_GCN_ALL_ROWS', '0') != '1'


_LABELS_OK = {}


def _head_classes(model):
    """Output width of a model that offers ``forward_head`` (models/gcn.py: loss and accuracy straight from the last
    aggregation's raw outputs, one kernel each way), else None."""
    if not hasattr(model, 'forward_head') or os.environ.get('DCR_FUSED_HEAD', '1') == '0':
        return None
    layers = getattr(model, 'layers', None)
    return int(getattr(layers[-1], 'out_channels', 0)) or None if layers is not None and len(layers) > 1 else None


def _labels_in_range(y, n_classes):
    """The head kernels read labels as int64 class ids in [0, n_classes): F.nll_loss raises on anything else except
    ignore_index = -100, which it leaves out of the mean — such labels keep the stock ops.  One host sync per label tensor
    (cached on its storage and version)."""
    if y.dtype != torch.int64 or not y.is_cuda or y.numel() == 0:
        return False
    key = (y.data_ptr(), y._version, tuple(y.shape), n_classes)
    hit = _LABELS_OK.get(key)
    if hit is None:
        if len(_LABELS_OK) >= 64:
            _LABELS_OK.clear()
        hit = _LABELS_OK[key] = bool(int(y.min()) >= 0 and int(y.max()) < n_classes)
    return hit


def train(model, optimizer, data):
    """One optimisation step on the training nodes; returns the loss value (training_loop.py:40-54)."""
    model.train()
    optimizer.zero_grad()
    mask = data.train_mask
    loss = None
    n_classes = _head_classes(model) if _takes_rows(model) else None
    if n_classes and _labels_in_range(data.y, n_classes):
        # (round 5) log_softmax + nll_loss on the training rows as one kernel behind the last aggregation: same loss, same gradient
        # up to float32 rounding; the captured epochs take the same route, so they stay bit-identical to this loop
        head = model.forward_head(data, rows_train=mask, y_train=data.y[mask].contiguous())
        loss = head[0] if head is not None else None
    if loss is not None:
        pass
    elif _takes_rows(model):
        loss = F.nll_loss(model(data, rows=mask), data.y[mask])
    else:
        log_probs = model(data)
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
    if not test and _takes_rows(model):
        mask = data['val_mask']
        n_classes = _head_classes(model)
        with torch.no_grad():
            head = None
            if n_classes and _labels_in_range(data.y, n_classes):
                head = model.forward_head(data, rows_eval=mask, y_eval=data.y[mask].contiguous())
            if head is not None:
                return {'val_acc': head[1].item() / mask.sum().item()}
            predicted = model(data, rows=mask).max(1)[1]
        return {'val_acc': predicted.eq(data.y[mask]).sum().item() / mask.sum().item()}
    with torch.no_grad():
        log_probs = model(data)
    splits = ('val', 'test') if test else ('val',)
    return {f'{split}_acc': _accuracy(log_probs, data.y, data[f'{split}_mask']) for split in splits}


def _join_side_streams():
    """A capture ends with every stream it forked joined: the side stream models/gcn.py draws the next epoch's dropout
    decisions on (joined by the first layer's backward already; this covers a step without one)."""
    from models import gcn as _gcn
    if hasattr(_gcn, 'join_dropout_ahead'):
        _gcn.join_dropout_ahead()


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
        self.rows = _takes_rows(model)
        # The one-kernel loss / accuracy (dcr_nll_picked_mean_*, dcr_count_argmax_equal_f32_dev) read the labels as int64
        # class ids in [0, C): F.nll_loss raises on anything else except ignore_index = -100, which it leaves out of the mean.
        # Checked once here (a host sync before anything is captured); labels they do not cover go the stock way.
        # Only the GRADIENT of the fused loss is bit-identical to F.nll_loss's (-1/m at the picked entries); the loss VALUE is
        # reduced in float64 and rounded once, so loss.item() may differ from the stock op in the last ulp.
        self._one = torch.ones((), dtype=torch.float32, device=data.x.device)   # d loss / d loss, kept (no fill launch per step)
        self.fused_labels = bool(data.y.dtype == torch.int64 and self.y_train.is_contiguous() and self.y_val.is_contiguous()
                                 and self.y_train.numel() > 0 and int(self.y_train.min()) >= 0 and int(self.y_val.min()) >= 0)
        self._n_classes_checked = None

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

    def _nll(self, log_probs):
        """F.nll_loss(log_probs[train_mask], y[train_mask]) (training_loop.py:51) with the same gradient, bit for bit (-1 / n at
        the selected entries): the stock kernel reduces the 100k selected rows in ONE workgroup (0.10 ms forward, 0.07
        backward at the bench shape), a gather and a mean are a few small multi-workgroup kernels."""
        if not self.rows:
            log_probs = log_probs.index_select(0, self.train_idx)
        if self.rows and _fused_ok(log_probs) and self._labels_fit(log_probs.shape[1]):
            return _PickedMean.apply(log_probs, self.y_train)
        return -log_probs.gather(1, self.y_train.unsqueeze(1)).mean()

    def _labels_fit(self, n_classes):
        """Labels are class ids below the model's output width (checked on the first, eager, epochs; cached)."""
        if not self.fused_labels:
            return False
        if self._n_classes_checked != n_classes:
            if torch.cuda.is_current_stream_capturing():
                return False   # (never reached: the eager warm-up epochs run the same code first)
            if max(int(self.y_train.max()), int(self.y_val.max())) >= n_classes:
                self.fused_labels = False   # F.nll_loss raises on such a target: let the stock op do so
                return False
            self._n_classes_checked = n_classes
        return True

    def _head(self, train, evaluate):
        """(loss, correct) from ``model.forward_head`` (one aggregation + one head kernel; models/gcn.py), or None when the model,
        the labels or the shapes ask for the log-probabilities and the separate kernels."""
        n_classes = _head_classes(self.model) if self.rows else None
        if not n_classes or not self._labels_fit(n_classes):
            return None
        return self.model.forward_head(self.data, rows_train=self.train_idx if train else None, y_train=self.y_train if train else None,
                                       rows_eval=self.val_idx if evaluate else None, y_eval=self.y_val if evaluate else None)

    def _train_step(self):
        head = self._head(True, False)
        if head is not None:
            loss = head[0]
        else:
            log_probs = self.model(self.data, rows=self.train_idx) if self.rows else self.model(self.data)
            loss = self._nll(log_probs)
        loss.backward(self._one if loss.dtype == torch.float32 else None)
        self.optimizer.step()
        return loss

    def _val_correct(self):
        with torch.no_grad():
            head = self._head(False, True)
            if head is not None:
                return head[1]
            if self.rows:
                lp = self.model(self.data, rows=self.val_idx)
                return _count_correct(lp, self.y_val) if _fused_ok(lp) and self._labels_fit(lp.shape[1]) else lp.max(1)[1].eq(self.y_val).sum()
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
                _join_side_streams()
            self.model.eval()
            self.eval_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.eval_graph, stream=self.stream):
                self.correct = self._val_correct()
        self.train_graph.replay()
        self.eval_graph.replay()
        return self.correct.item() / self.n_val


class LaggedGraphedEpoch(GraphedEpoch):
    """ONE captured HIP graph per epoch for models that offer ``forward_pair`` (models/gcn.py).

    The reference's loop is train(e), evaluate(e), train(e+1), ... (training_loop.py:25-26), and evaluate(e) and the forward
    of train(e+1) run on the same weights.  So the graph of step e+1 is: snapshot the weights, ``forward_pair`` (training
    log-probabilities AND the evaluation log-probabilities of the weights as they stand, i.e. of epoch e), loss, backward,
    optimiser step.  A replay therefore performs training step e+1 and delivers the validation accuracy of epoch e: the
    first layer's GEMM and every aggregation are paid once for both (1.1 of the 5.3 ms of an epoch at the 1M-node
    shape).  ``training_loop`` consumes the accuracies one step late: the checkpoint of epoch e is taken from the snapshot,
    and a step that turns out to follow the stopping epoch is discarded with the rest when the best weights are restored,
    so the returned model and every accuracy are what the plain loop gives.

    ``step()`` = one replay; ``accuracy_before_step()`` = validation accuracy of the weights the last step started from;
    ``snapshot()`` = those weights; ``accuracy_now()`` = a plain evaluation graph (used once, after the last step)."""

    def __init__(self, model, optimizer, data):
        super().__init__(model, optimizer, data)
        self.graph = None
        self.prev = None

    @staticmethod
    def supported(model, optimizer, data):
        return (hasattr(model, 'forward_pair') and os.environ.get('DCR_LAGGED_EVAL', '1') != '0'
                and GraphedEpoch.supported(model, optimizer, data))

    def _fused_step(self):
        torch._foreach_copy_(self.prev, list(self.model.state_dict().values()))   # (one launch for the snapshot)
        head = self._head(True, True)
        if head is not None:
            loss, correct_prev = head
            loss.backward(self._one)
            self.optimizer.step()
            return correct_prev
        if self.rows:
            lp_train, lp_eval = self.model.forward_pair(self.data, rows_train=self.train_idx, rows_eval=self.val_idx)
            correct_prev = (_count_correct(lp_eval, self.y_val) if _fused_ok(lp_eval) and self._labels_fit(lp_eval.shape[1])
                            else lp_eval.max(1)[1].eq(self.y_val).sum())
        else:
            lp_train, lp_eval = self.model.forward_pair(self.data)
            correct_prev = lp_eval.index_select(0, self.val_idx).max(1)[1].eq(self.y_val).sum()
        loss = self._nll(lp_train)
        loss.backward()
        self.optimizer.step()
        return correct_prev

    def step(self):
        cur = torch.cuda.current_stream(self.data.x.device)
        if self.graph is None:
            self.prev = [v.detach().clone() for v in self.model.state_dict().values()]
            self.model.train()
            self.optimizer.zero_grad(set_to_none=True)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):  # one eager run of the fused step's own kernels before the capture
                self.optimizer.zero_grad(set_to_none=True)
                self._warm = self._fused_step()
            cur.wait_stream(self.stream)
            self.optimizer.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
            self._pending_capture = True
            return self._warm
        if self._pending_capture:
            self._pending_capture = False
            self.model.train()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.correct_prev = self._fused_step()
                _join_side_streams()
        self.graph.replay()
        return self.correct_prev

    def snapshot(self):
        return {k: v.clone() for k, v in zip(self.model.state_dict().keys(), self.prev)}

    def accuracy_now(self):
        if self.eval_graph is None:
            self.model.eval()
            cur = torch.cuda.current_stream(self.data.x.device)
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                self._val_correct()
            cur.wait_stream(self.stream)
            self.eval_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.eval_graph, stream=self.stream):
                self.correct = self._val_correct()
            self.model.train()
        self.eval_graph.replay()
        return self.correct.item() / self.n_val

    def __call__(self):
        """For drivers that only want epochs done (bench.py): one training step + the evaluation that travels with it;
        returns the validation accuracy of the weights the step started from."""
        return self.step().item() / self.n_val


def make_epoch(model, optimizer, data, lagged=False):
    """A callable that runs one epoch (training step + validation forward) and returns the validation accuracy:
    graph replay where supported, the eager ``train`` / ``evaluate`` pair otherwise.  ``lagged=True`` (throughput
    drivers): the one-graph epoch whose accuracy is that of the previous step's weights (``LaggedGraphedEpoch``)."""
    if lagged and LaggedGraphedEpoch.supported(model, optimizer, data):
        return LaggedGraphedEpoch(model, optimizer, data)
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

    def account(val_acc, weights):
        """training_loop.py:27-33 for one epoch; True: stop."""
        nonlocal best_acc, best_weights, since_best
        if val_acc >= best_acc:  # ties advance the checkpoint, as in the reference
            best_acc, since_best = val_acc, 0
            best_weights = weights()
        else:
            since_best += 1
        return since_best >= patience

    if epochs > 0 and LaggedGraphedEpoch.supported(model, optimizer, data):
        # one graph per epoch; the accuracy of epoch e arrives with the replay of step e + 1 (see LaggedGraphedEpoch)
        lag = LaggedGraphedEpoch(model, optimizer, data)
        stopped = False
        lag.step()                                           # training step of epoch 0 (its accuracy output: of the initial weights, unused)
        for e in range(1, epochs):
            correct = lag.step()                             # training step e, accuracy of epoch e - 1
            if account(correct.item() / lag.n_val, lag.snapshot):
                stopped = True                               # the reference stops after epoch e - 1: step e is discarded below
                break
        if not stopped:
            account(lag.accuracy_now(), lambda: copy.deepcopy(model.state_dict()))
        model.load_state_dict(best_weights)
        return model
    epoch = make_epoch(model, optimizer, data)
    for _ in range(epochs):
        if account(epoch(), lambda: copy.deepcopy(model.state_dict())):
            break
    model.load_state_dict(best_weights)
    return model
