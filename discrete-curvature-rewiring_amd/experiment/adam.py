"""Adam as ONE launch per step (csrc/dcr_gcn.hip, dcr_adam_step_f32_dev): the reference's optimiser wiring
(experiment/save_models.py:78-82: ``Adam`` over two parameter groups, weight decay on the first layer only, L2 in the gradient)
for the handful of small tensors of a GCN.  torch's own implementations take two to three launches per parameter group
(``fused=True``) or ~45 per step (the default); at a citation-sized graph the optimiser was a fifth of a captured epoch.

Same update rule as ``torch.optim.Adam`` (amsgrad off, maximize off), float32, another order of roundings: weights trained with
it are comparable to the stock optimiser's to rounding, not bit for bit — ``experiment.save_models.make_adam`` selects it with
``DCR_FUSED_ADAM=2`` (``bench.py`` does and says so); the default stays the stock implementation."""
import ctypes

import torch


class OneLaunchAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid Adam hyper-parameter')
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True)
        super().__init__(params, defaults)
        first = self.param_groups[0]
        for g in self.param_groups:
            if (g['lr'], tuple(g['betas']), g['eps']) != (first['lr'], tuple(first['betas']), first['eps']):
                raise ValueError('OneLaunchAdam: the parameter groups may differ in weight_decay only (one launch, one step size)')
        self._step = None      # float32 [1] on the device: torch's capturable Adam keeps its counters the same way
        self._ticket = None

    @torch.no_grad()
    def step(self, closure=None):
        from dcr import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps, wds = [], []
        for g in self.param_groups:
            for p in g['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError('OneLaunchAdam: contiguous float32 parameters on the GPU expected')
                st = self.state[p]
                if not st:
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                ps.append(p)
                wds.append(float(g['weight_decay']))
        if not ps:
            return loss
        dev = ps[0].device
        if self._step is None:
            self._step = torch.zeros(1, dtype=torch.float32, device=dev)
            self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        g0 = self.param_groups[0]
        stream = torch.cuda.current_stream(dev).cuda_stream
        VP = ctypes.c_void_p
        for lo in range(0, len(ps), 8):
            part, wd = ps[lo:lo + 8], wds[lo:lo + 8]
            n = len(part)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in part]
            arr = lambda ts: (VP * n)(*[t.data_ptr() for t in ts])
            # (several launches of one step must not advance the counter more than once: the kernel advances it, so later parts
            #  of a step get a counter rewound by one first — models of this package have four tensors: one launch)
            if lo:
                self._step.sub_(1)
            _lib.check(_lib.lib().dcr_adam_step_f32_dev(
                n, arr(part), arr(grads), arr([self.state[p]['exp_avg'] for p in part]), arr([self.state[p]['exp_avg_sq'] for p in part]),
                (ctypes.c_int64 * n)(*[p.numel() for p in part]), (ctypes.c_float * n)(*wd), float(g0['lr']), float(g0['betas'][0]),
                float(g0['betas'][1]), float(g0['eps']), self._step.data_ptr(), self._ticket.data_ptr(), VP(stream)))
        return loss
