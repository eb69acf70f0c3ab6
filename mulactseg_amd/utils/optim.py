"""AdamW on the package's one-launch kernel (``csrc/optim.hip``) -- the optimizer of the reference's ``trainer/base.py:64-66``
(``optim.AdamW(params=[backbone @ lr, classifier @ cls_lr_scale * lr], weight_decay=wd)``, torch 1.11's single-tensor update rule).

A subclass of ``torch.optim.AdamW``: parameter groups, ``state_dict`` / ``load_state_dict`` (the ``opt_state_dict`` of the reference's
checkpoints, ``trainer/base.py:281-294``), learning-rate schedulers and the optimizer step hooks are torch's; only ``step`` is
replaced.  Per step: ONE launch over a device-resident job table (rebuilt only when a gradient buffer moved), the groups' learning
rates as kernel arguments, the step count on the device.  ``optimizer.found_inf`` (the GradScaler protocol the trainers use for the
stream-K give-up word, ``trainer/base.py:guard_optimizer_step``): a non-zero device scalar leaves parameters, moments and the step
count untouched -- no host read."""
import ctypes

import torch

from .. import _lib


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        lib = _lib.load()
        if len(self.param_groups) > int(lib.mas_adamw_max_groups()):
            raise ValueError("at most %d parameter groups" % int(lib.mas_adamw_max_groups()))
        for g in self.param_groups:
            if g.get('amsgrad') or g.get('maximize'):
                raise NotImplementedError("amsgrad / maximize are outside the reference's configuration")
            g['fused'] = True           # (read by trainer/base.py:guard_optimizer_step: this optimizer honours `found_inf` on the device)
        self._step_dev = None           # device float: steps taken so far
        self._table = None              # (key, device job table, njobs, nblocks)
        self._rec = int(lib.mas_adamw_job_bytes())

    def _hyper(self):
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if (g['betas'], g['eps'], g['weight_decay']) != (g0['betas'], g0['eps'], g0['weight_decay']):
                raise NotImplementedError("parameter groups may differ in their learning rate only")
        return g0['betas'][0], g0['betas'][1], g0['eps'], g0['weight_decay']

    def _ensure_state(self, p, dev):
        st = self.state[p]
        if 'exp_avg' not in st:
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if self._step_dev is None:
            # a loaded state carries per-parameter counts (torch's layout): they are all the same number -- one device scalar from here on
            old = st.get('step')
            self._step_dev = torch.full((), float(old) if old is not None else 0.0, dtype=torch.float32, device=dev)
        st['step'] = self._step_dev     # (every parameter shares it: they step together; state_dict() writes it once per parameter)
        return st

    def _upload(self, raw, dev):
        """The job table to the device through a page-locked slot: the copy is only ENQUEUED (a copy from pageable memory would hold the
        host until the backward pass in front of it has drained).  A slot is reused after the event recorded behind its copy."""
        ring = getattr(self, '_ring', None)
        if ring is None:
            ring = self._ring = {'bufs': [None] * 4, 'events': [None] * 4, 'k': 0}
        k = ring['k']
        ring['k'] = (k + 1) % len(ring['bufs'])
        if ring['events'][k] is not None:
            ring['events'][k].synchronize()
        n = len(raw)
        if ring['bufs'][k] is None or ring['bufs'][k].numel() < n:
            ring['bufs'][k] = torch.empty(max(n, 1 << 16), dtype=torch.uint8).pin_memory()
        host = ring['bufs'][k][:n]
        host.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        table = torch.empty(n, dtype=torch.uint8, device=dev)
        table.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        ring['events'][k] = ev
        return table

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        work = []
        dev = None
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                    raise TypeError("FusedAdamW takes f32 parameters and dense f32 gradients on the GPU")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError("FusedAdamW needs contiguous parameters and gradients")
                dev = p.device if dev is None else dev
                st = self._ensure_state(p, p.device)
                work.append((p, p.grad, st['exp_avg'], st['exp_avg_sq'], gi))
        if not work:
            return loss
        key = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), gi) for p, g, m, v, gi in work)
        if self._table is None or self._table[0] != key:
            host = ctypes.create_string_buffer(self._rec * len(work))
            base = ctypes.addressof(host)
            first = 0
            for i, (p, g, m, v, gi) in enumerate(work):
                n = lib.mas_adamw_job(base + i * self._rec, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), gi, first)
                if n == 0:
                    raise _lib.MulActSegHipError("mas_adamw_job rejected parameter %d" % i)
                first += n
            self._table = (key, self._upload(host.raw, dev), len(work), first)
        _, table, njobs, nblocks = self._table
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g['lr']) for g in self.param_groups])
        b1, b2, eps, wd = self._hyper()
        skip = getattr(self, 'found_inf', None)
        if skip is not None:
            skip = skip.to(device=dev, dtype=torch.float32).reshape(())
        with torch.cuda.device(dev):
            _lib.check(lib.mas_adamw_multi(table.data_ptr(), njobs, nblocks, ctypes.cast(lrs, ctypes.c_void_p), len(self.param_groups),
                                           b1, b2, eps, wd, self._step_dev.data_ptr(), skip.data_ptr() if skip is not None else None,
                                           torch.cuda.current_stream(dev).cuda_stream), "mas_adamw_multi")
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_dev, self._table = None, None
        for g in self.param_groups:
            g['fused'] = True
