"""Mirror of deeplio/models/optimizer.py:4-16 (`create_optimizer(params, cfg, args)`), built for
MI355X: all parameters of all groups are re-homed into ONE flat fp32 buffer (grads into a
second one), so the optimizer step is a single HBM-bound kernel over 41-65 M floats and the
data-parallel gradient exchange is a single large RCCL all-reduce over xGMI
(deeplio_amd.dist).  Weight decay is L2 added to the gradient, as torch.optim does."""
import os

import torch

from . import ops

ALIGN = 16  # floats (64 B): every parameter view stays float4-aligned


def _join():
    """gradients are accumulated on auxiliary HIP streams (functional.aux_stream): the stream the
    step runs on must wait for them"""
    from .functional import join_aux_streams
    join_aux_streams()


class FlatOptimizer:
    def __init__(self, params, lr, weight_decay):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{'params': groups}]
        self.param_groups = []
        plist = []
        for g in groups:
            ps = [p for p in g['params']]
            self.param_groups.append({'params': ps, 'lr': g.get('lr', lr),
                                      'weight_decay': g.get('weight_decay', weight_decay)})
            plist += ps
        seen, uniq = set(), []
        for p in plist:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        if not uniq:
            raise ValueError("optimizer got an empty parameter list")
        dev = uniq[0].device
        if dev.type != "cuda":
            raise RuntimeError("deeplio_amd optimizers need parameters on the HIP device")
        offs, total = [], 0
        for p in uniq:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(uniq, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.offsets = offs
        self.step_count = 0
        self.grad_scale = 1.0
        self._early = None           # (lo, event): [lo:] of this step was already updated by step_early()
        self.early_blocks = int(os.environ.get("DLIO_EARLY_STEP_BLOCKS", "256"))

    def _apply(self, lo, hi, count):
        """the update rule over elements [lo:hi) of the flat buffers (one launch); `count` = this step's number"""
        raise NotImplementedError

    @torch.no_grad()
    def step(self):
        _join()
        self._hyper()
        self.step_count += 1
        hi = self.flat.numel()
        if self._early is not None:
            hi, ev = self._early
            self._early = None
            torch.cuda.current_stream().wait_event(ev)
        if hi > 0:
            self._apply(0, hi, self.step_count)

    @torch.no_grad()
    def step_early(self, lo, stream=None):
        """update elements [lo:] NOW -- their gradients are final although backward is still running (TrainStep: everything
        behind the feature nets, 87 % of the headline model, is final when backward reaches the fusion output) -- on `stream`
        (which must already be ordered behind the kernels that wrote those gradients); the following step() then only
        updates [:lo] and waits for this launch.  The end of the step loses the 165 MB x 7 sweep's largest part: it runs
        under the encoder backward instead of behind it."""
        lo = int(lo)
        if self._early is not None or lo <= 0 or lo >= self.flat.numel() or lo % 4:
            return False
        self._hyper()
        # (the weight-layout cache's epoch is bumped by the step() that ends this training step, not here: the range holds
        #  Linear / RNN parameters only -- no convolution layouts -- and a bump now would make any on-demand layout fetch of
        #  the backward pass still running rebuild every layout)
        epoch = ops._PREP.epoch
        # background work: a small grid (one workgroup per CU) leaves the chip to the critical chain it runs beside
        ops.optim_set_max_blocks(self.early_blocks)
        if stream is None:
            self._apply(lo, self.flat.numel(), self.step_count + 1)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        else:
            from .functional import on_stream
            with on_stream(stream):
                self._apply(lo, self.flat.numel(), self.step_count + 1)
                ev = torch.cuda.Event()
                ev.record(stream)
        ops.optim_set_max_blocks(0)
        ops._PREP.epoch = epoch
        self._early = (lo, ev)
        return True

    def set_overwritten(self, params):
        """`params` (a contiguous run of this optimizer's parameters, in order): their gradient slots are OVERWRITTEN by their
        producer in every backward pass (functional.LstmStackFn writes a layer's weight gradients in one launch, exactly once)
        or never written at all (a direction that is not run: stays zero) -- zero_grad() then skips them: no fill over, and no
        read-modify-write into, the 143 MB of the odometry LSTM's gradients.  Only for a loop that runs exactly one backward
        pass per step (TrainStep): gradient accumulation over several passes needs the default.  None / empty: off."""
        self._skip = None
        for p in self.params:
            if hasattr(p, "_dlio_grad_overwrite"):
                del p._dlio_grad_overwrite
        ps = list(params or [])
        if not ps:
            return
        index = {id(p): i for i, p in enumerate(self.params)}
        idx = sorted(index[id(p)] for p in ps)
        if idx != list(range(idx[0], idx[0] + len(idx))):
            raise ValueError("set_overwritten needs a contiguous run of parameters")
        lo = self.offsets[idx[0]]
        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else self.flat.numel()
        self.grad[lo:hi].zero_()                 # (the slots nobody ever writes must hold zeros from now on)
        for p in ps:
            p._dlio_grad_overwrite = True
        self._skip = (lo, hi)

    def zero_grad(self, set_to_none=False):
        skip = getattr(self, "_skip", None)
        if skip is None:
            self.grad.zero_()
        else:
            if skip[0] > 0:
                self.grad[:skip[0]].zero_()
            if skip[1] < self.grad.numel():
                self.grad[skip[1]:].zero_()
        for p, o in zip(self.params, self.offsets):   # keep .grad views attached
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def _hyper(self):
        """one (lr, weight decay) for the whole flat buffer: the reference builds every optimizer with
        a single lr / weight decay for its two groups (trainer.py:56-58); per-group values would need
        one launch per group and are rejected instead of silently merged"""
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if g['lr'] != g0['lr'] or g['weight_decay'] != g0['weight_decay']:
                raise ValueError("flat optimizer needs identical hyper-parameters in all groups "
                                 "(the reference uses one lr / weight-decay for model and criterion)")
        return g0['lr'], g0['weight_decay']

    def grad_norm(self):
        """calc_grad_norm (trainer.py:481-486) without per-tensor launches"""
        return float(ops.sumsq(self.grad).sqrt().item())

    def state_dict(self):
        """the layout torch.optim.<same class>.state_dict() has ({'state': {i: {...}}, 'param_groups': [...]}): the
        reference's Trainer saves this entry (trainer.py:161) and loads it back (trainer.py:108), so a checkpoint
        written through the overlay resumes under the reference's own torch.optim optimizer and vice versa"""
        from .checkpoint import optimizer_to_torch_state
        return optimizer_to_torch_state(self)

    def load_state_dict(self, sd):
        """torch.optim layout (what the reference writes), or the flat layout this class wrote before round 3
        ({'step', 'state': {buffer name: flat tensor}, 'param_groups'})"""
        if 'step' in sd and all(isinstance(k, str) for k in sd.get('state', {})):
            self.step_count = int(sd['step'])
            for k, v in sd['state'].items():
                self._state()[k].copy_(v)
            for g, s in zip(self.param_groups, sd['param_groups']):
                g.update({k: v for k, v in s.items() if k != 'params'})
            return
        from .checkpoint import optimizer_from_torch_state
        optimizer_from_torch_state(self, sd)


class Adam(FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.):
        super().__init__(params, lr, weight_decay)
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)

    def _state(self):
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}

    def _apply(self, lo, hi, count):
        lr, wd = self._hyper()
        ops.adam_step(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], lr, self.betas[0],
                      self.betas[1], self.eps, wd, count, self.grad_scale)


class SGD(FlatOptimizer):
    def __init__(self, params, lr, momentum=0., weight_decay=0.):
        super().__init__(params, lr, weight_decay)
        self.momentum = momentum
        self.buf = torch.zeros_like(self.flat)

    def _state(self):
        return {'momentum_buffer': self.buf}

    def _apply(self, lo, hi, count):
        lr, wd = self._hyper()
        ops.sgd_step(self.flat[lo:hi], self.grad[lo:hi], self.buf[lo:hi], lr, self.momentum, wd, count, self.grad_scale)


class RMSprop(FlatOptimizer):
    """torch.optim.RMSprop (optimizer.py:12-13) over the flat buffers"""

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0., momentum=0., centered=False):
        super().__init__(params, lr, weight_decay)
        self.alpha, self.eps, self.momentum, self.centered = alpha, eps, momentum, centered
        self.square_avg = torch.zeros_like(self.flat)
        self.buf = torch.zeros_like(self.flat) if momentum != 0. else None
        self.grad_avg = torch.zeros_like(self.flat) if centered else None

    def _state(self):
        st = {'square_avg': self.square_avg}
        if self.buf is not None:
            st['momentum_buffer'] = self.buf
        if self.grad_avg is not None:
            st['grad_avg'] = self.grad_avg
        return st

    def _apply(self, lo, hi, count):
        lr, wd = self._hyper()
        ops.rmsprop_step(self.flat[lo:hi], self.grad[lo:hi], self.square_avg[lo:hi],
                         None if self.buf is None else self.buf[lo:hi], None if self.grad_avg is None else self.grad_avg[lo:hi],
                         lr, self.alpha, self.eps, wd, self.momentum, self.grad_scale)


class Adadelta(FlatOptimizer):
    """torch.optim.Adadelta (optimizer.py:14-15) over the flat buffers"""

    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6, weight_decay=0.):
        super().__init__(params, lr, weight_decay)
        self.rho, self.eps = rho, eps
        self.square_avg = torch.zeros_like(self.flat)
        self.acc_delta = torch.zeros_like(self.flat)

    def _state(self):
        return {'square_avg': self.square_avg, 'acc_delta': self.acc_delta}

    def _apply(self, lo, hi, count):
        lr, wd = self._hyper()
        ops.adadelta_step(self.flat[lo:hi], self.grad[lo:hi], self.square_avg[lo:hi], self.acc_delta[lo:hi], lr, self.rho,
                          self.eps, wd, self.grad_scale)


def create_optimizer(params, cfg, args, **kwargs):
    optim_type = cfg['optimizer'].lower()
    if optim_type == 'sgd':
        return SGD(params, lr=args.lr, weight_decay=args.weight_decay, momentum=args.momentum, **kwargs)
    if optim_type == 'adam':
        return Adam(params, lr=args.lr, weight_decay=args.weight_decay, **kwargs)
    if optim_type == 'rmsprop':
        return RMSprop(params, lr=args.lr, weight_decay=args.weight_decay, **kwargs)
    if optim_type == 'adadelta':
        return Adadelta(params, lr=args.lr, weight_decay=args.weight_decay, **kwargs)
    raise ValueError("Optimizer {} not supported!".format(optim_type))
