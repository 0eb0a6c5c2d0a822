"""The per-iteration body of Trainer.train (trainer.py:213-281) on the HIP path: forward,
SE(3) chain, loss, backward, (data-parallel gradient exchange), optimizer step -- with the
reference's ten NaN/Inf host syncs per step replaced by device-side flags that the caller
reads when it wants to (check())."""
import gc
import os
import types

import torch

from . import functional as Fh
from . import ops
from .losses import get_loss_function
from .misc import build_config_container
from .nets import get_model
from .optimizer import create_optimizer
from .se3 import se3_to_SE3


_GC_ARMED = [0]       # TrainSteps that froze the collector's generations / switched it to manual (release_gc)
_GC_MANUAL = [0]

class TrainStep:
    def __init__(self, cfg, input_shape, device, batch_size, lr=1e-3, weight_decay=1e-4, momentum=0.9,
                 max_glob_seq=2, grad_sync=None):
        self.cfg, self.device = cfg, torch.device(device)
        self.args = types.SimpleNamespace(device=str(device), batch_size=batch_size, lr=lr,
                                          weight_decay=weight_decay, momentum=momentum)
        build_config_container(cfg, self.args)
        Fh.assign_streams(self.device)              # the heavy streams of the step on hardware queues of their own
        self.model = get_model(input_shape, cfg, self.device)
        self.criterion = get_loss_function(cfg, self.device)
        self.optimizer = create_optimizer([{'params': self.model.parameters()},
                                           {'params': self.criterion.parameters()}], cfg, self.args)
        self.max_glob_seq = max_glob_seq            # trainer.py:42
        self.fused_tail = True                      # functional.PoseTailFn instead of se3_to_SE3 + slices + criterion
        self.flags = torch.zeros(2, dtype=torch.int32, device=self.device)   # [nonfinite, se3 status]
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.grad_sync = None
        # Host-side hygiene: a step creates ~1e4 short-lived Python objects (tensors, autograd contexts), so
        # CPython's cyclic collector runs many times per step and its full collections walk every object
        # of the model; in the host-bound serial middle of the step each pause leaves the GPU idle
        # (measured: 31.4 vs 28.6 ms/step).  'freeze' (default): move everything that exists after
        # construction into the permanent generation (gc.freeze) so collections only see the step's own
        # objects; 'manual': collector off, one explicit collection every `gc_every` steps; 'default':
        # leave the interpreter alone.
        self.gc_mode = os.environ.get("DLIO_GC", "freeze")
        self.gc_every = int(os.environ.get("DLIO_GC_EVERY", "100"))
        self._gc_armed = False
        self._steps = 0
        # Device-side error words (non-finite output, det != 1, a cooperative BatchNorm launch at its spin limit) are
        # polled WITHOUT a host sync: every `check_every` steps they are copied asynchronously into pinned memory and the
        # copy issued `check_every` steps earlier (long complete) is inspected -- a bad step raises at most 2 x check_every
        # steps later instead of training on until somebody calls check().  0 switches the polling off.
        self.check_every = int(os.environ.get("DLIO_CHECK_EVERY", "8"))
        # backward on the calling thread instead of the engine's device thread (no hand-over, no second thread taking turns at
        # the interpreter lock with the one that issues the launches): see DESIGN 9 for the measured host time
        self.autograd_inline = os.environ.get("DLIO_AUTOGRAD_INLINE", "1") != "0"
        self._poll_pending = None
        self._poll_host = None
        # the optimizer step over the tail bucket issued from inside backward (see _tail_ready); 0: one sweep at the end
        self.early_tail_step = os.environ.get("DLIO_EARLY_TAIL_STEP", "1") != "0"
        # (the gradient buffer's fill beside the forward pass instead of in the serial middle: median 18.00 against 17.96 ms,
        #  eight alternations -- no gain: off)
        self.zero_grad_early = os.environ.get("DLIO_ZERO_GRAD_EARLY", "0") != "0"
        self._tail_lo = self.tail_offset()
        if self._tail_lo is not None and (self._tail_lo <= 0 or self._tail_lo % 4):
            self._tail_lo = None
        self.model.tail_grads_ready = self._tail_ready
        # DLIO_GRAD_OVERWRITE=1: the odometry LSTM's gradient slots (143 of the 165 MB) are written once per step by
        # functional.LstmStackFn and neither zero-filled nor read back (FlatOptimizer.set_overwritten).  Bit-identical training
        # (test_lstm_gradient_slots_overwritten_instead_of_zeroed), 190 MB less traffic per step -- and no measurable gain
        # (median 18.35 against 18.38 ms, ten alternations): off by default, the plain fill + accumulate contract stays
        self.overwrite_lstm_grads = os.environ.get("DLIO_GRAD_OVERWRITE", "0") != "0"
        self._set_overwritten()
        self.model.train()
        if grad_sync is not None:
            self.set_grad_sync(grad_sync)
        self.criterion.train()

    def tail_offset(self):
        """flat-buffer offset of the first parameter behind the feature nets (odometry net, else the
        heads): everything from there on is final when backward reaches the fusion output"""
        m = self.model
        first = next((p for n in (m.odom_feat_net, m.fc_pos) if n is not None for p in n.parameters()), None)
        if first is None:
            return None
        index = {id(p): i for i, p in enumerate(self.optimizer.params)}
        return self.optimizer.offsets[index[id(first)]]

    def _set_overwritten(self):
        from .nets import OdomFeatRNN
        net = self.model.odom_feat_net
        ok = (self.overwrite_lstm_grads and isinstance(net, OdomFeatRNN) and hasattr(self.optimizer, "set_overwritten")
              and self.device.type == "cuda" and net.rnn.mode == "lstm" and Fh._LSTM_LAYER[0])
        if ok:
            r = net.rnn
            D = 2 if r.bidirectional else 1
            B = self.args.batch_size
            S = self.model.seq_size
            ok = (B <= 8 and all(ops.lstm_layer_ok(S, B, r.input_size if l == 0 else D * r.hidden_size, r.hidden_size, D)
                                 for l in range(r.num_layers))
                  and all(p.requires_grad for p in r.parameters()))
        self.optimizer.set_overwritten(list(net.rnn.parameters()) if ok else None)

    def _tail_ready(self):
        """autograd hook on the fusion output's gradient: everything behind the feature nets (odometry net, heads, loss weights:
        [tail_offset():] of the flat buffers, 87 % of the headline model) has its final gradient.  Data parallel: start that
        bucket's all-reduce; then run the optimizer over it NOW on the 'comm' stream (beside a weight-gradient companion,
        functional.assign_streams) -- under the encoder backward instead of at the end of the step."""
        sync = self.grad_sync
        dp = sync is not None and sync.world > 1
        if dp and sync.tail_lo is not None:
            sync.reduce_tail_async()
        if not self.early_tail_step or self._tail_lo is None or not hasattr(self.optimizer, "step_early"):
            return
        if dp and not isinstance(getattr(sync, "_tail_work", None), torch.cuda.Event):
            return                     # (the bucket is not being reduced on the comm stream: the whole step at the end)
        if self.device.type != "cuda":
            return
        overlap = getattr(self.model, "side_stream", True)
        if overlap:
            comm = Fh.aux_stream(self.device, "comm")
            cur = Fh.current_stream_obj()
            ws = Fh.wgrad_stream_of(cur)
            if ws is not None:
                comm.wait_stream(ws)   # (the tail's weight gradients were forked onto the companion stream: the sweep waits for
            comm.wait_stream(cur)      #  them, the chain on `cur` does not)
            self.optimizer.step_early(self._tail_lo, comm)
        else:
            self.optimizer.step_early(self._tail_lo, None)

    def set_grad_sync(self, sync):
        """data parallel: gradient exchange in two buckets, the tail one overlapped with backward"""
        self.grad_sync = sync
        if sync is not None and sync.world > 1 and "DLIO_CHECK_EVERY" not in os.environ:
            self.check_every = 1          # beside RCCL's kernels the cooperative launches are least certain of their partners
        if sync is not None and sync.world > 1:
            sync.set_tail(self.tail_offset())

    def _manage_gc(self):
        if self.gc_mode == "default":
            return
        if not self._gc_armed:
            self._gc_armed = True
            gc.collect()
            gc.freeze()
            _GC_ARMED[0] += 1             # gc.freeze() / disable() are process-global: counted over all live TrainSteps
            if self.gc_mode == "manual":
                _GC_MANUAL[0] += 1
                gc.disable()
        elif self.gc_mode == "manual" and self._steps % self.gc_every == 0:
            gc.collect()

    def release_gc(self):
        """give the interpreter's collector back (end of training): the objects frozen by the first step return to the
        collector's generations (gc.freeze() is process-global -- an embedding program, e.g. the reference's Trainer
        building several steps, must not keep them exempt for good)"""
        if self._gc_armed:
            _GC_ARMED[0] -= 1
            if _GC_ARMED[0] <= 0:         # the last armed step releases: another TrainStep (train + validate) keeps its freeze
                _GC_ARMED[0] = 0
                gc.unfreeze()
            if self.gc_mode == "manual":
                _GC_MANUAL[0] -= 1
                if _GC_MANUAL[0] <= 0:
                    _GC_MANUAL[0] = 0
                    gc.enable()
        self._gc_armed = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release_gc()
        return False

    def _tail(self, feats, gts_f2f, gts_f2g, hook=True):
        """features -> loss: the model's last layers, NaN/Inf flags, SE(3) chain, criterion"""
        pred_f2f_t, pred_f2f_w = self.model.forward_tail(feats, grads_ready_hook=hook)
        crit, lt = self.criterion, self.criterion.loss_Types
        g0, g1 = 1, min(self.max_glob_seq + 1, pred_f2f_t.shape[1])
        if (self.fused_tail and pred_f2f_t.is_cuda and g0 < g1 and (lt[0] or lt[1]) and hasattr(crit, "mode")
                and tuple(gts_f2f.shape) == tuple(pred_f2f_t.shape[:2]) + (6,)
                and tuple(gts_f2g.shape) == tuple(pred_f2f_t.shape[:2]) + (7,)):
            # non-finite check, SE(3) chain and criterion as one tape node (two launches each way; same kernels as below)
            return Fh.PoseTailFn.apply(getattr(crit, "sx", None), getattr(crit, "sq", None), float(getattr(crit, "beta", 0.)),
                                       crit.mode | (2 if crit.rotation == 'geodesic' else 0),
                                       (1 if lt[0] else 0) | (2 if lt[1] else 0), pred_f2f_t, pred_f2f_w, gts_f2f, gts_f2g,
                                       g0, g1, 0, self.flags[1:2], self.flags[0:1])
        gt_f2f_t, gt_f2f_w = gts_f2f[:, :, 0:3], gts_f2f[:, :, 3:]
        gt_f2g_p, gt_f2g_q = gts_f2g[:, :, 0:3], gts_f2g[:, :, 3:7]
        ops.nonfinite_flag(pred_f2f_t, self.flags[0:1])        # trainer.py:240-243, no host sync
        ops.nonfinite_flag(pred_f2f_w, self.flags[0:1])
        pred_f2g_p, pred_f2g_q = se3_to_SE3(pred_f2f_t, pred_f2f_w, status=self.flags[1:2])
        lt = self.criterion.loss_Types
        if lt[0] and not lt[1]:
            pred_f2g_p, pred_f2g_q = pred_f2g_p.detach(), pred_f2g_q.detach()
        elif lt[1] and not lt[0]:
            pred_f2f_t, pred_f2f_w = pred_f2f_t.detach(), pred_f2f_w.detach()
        sl = slice(1, self.max_glob_seq + 1)
        return self.criterion(pred_f2f_t, pred_f2f_w, pred_f2g_p[:, sl, :], pred_f2g_q[:, sl, :],
                              gt_f2f_t, gt_f2f_w, gt_f2g_p[:, sl, :], gt_f2g_q[:, sl, :])

    def step(self, imgs, normals, imus, gts_f2f, gts_f2g):
        self._steps += 1
        self._manage_gc()
        Fh.lazy_clear()            # (entries a failed backward pass may have left behind)
        if getattr(self.optimizer, "_early", None) is not None:
            # a pass that raised between its early tail sweep and optimizer.step(): that sweep stands (the caller restores its
            # last good state after an error), its marker must not make THIS step skip the tail
            self.optimizer._early = None
        zero_ev = None
        if self.zero_grad_early and self.device.type == "cuda" and getattr(self.model, "side_stream", True):
            # the 165 MB fill of the flat gradient buffer runs on the `comm` stream beside the forward pass (its hardware queue is
            # idle until backward) instead of between the loss and backward(), in the step's serial middle
            comm = Fh.aux_stream(self.device, "comm")
            cur = Fh.current_stream_obj()
            comm.wait_stream(cur)                  # behind the previous step's optimizer sweep
            with Fh.on_stream(comm):
                self.optimizer.zero_grad()
                zero_ev = torch.cuda.Event()
                zero_ev.record(comm)
        loss = self._tail(self.model.forward_features([[imgs, normals], imus]), gts_f2f, gts_f2g)
        if zero_ev is None:
            self.optimizer.zero_grad()
        else:
            Fh.current_stream_obj().wait_event(zero_ev)
        if self.autograd_inline:
            with torch.autograd.set_multithreading_enabled(False):
                loss.backward(self._one)           # (an explicit d loss / d loss: no fill launch per step)
        else:
            loss.backward(self._one)
        if self.grad_sync is not None:
            self.grad_sync.all_reduce_grads()
        self.optimizer.step()
        if self.check_every > 0 and self._steps % self.check_every == 0:
            self._poll()
        return loss.detach()

    def _poll(self):
        """inspect the error words copied at the previous poll, then start the next asynchronous copy"""
        pend = self._poll_pending
        self._poll_pending = None
        if pend is not None:
            host, ev = pend
            ev.synchronize()               # recorded check_every steps ago: complete unless the host runs that far ahead
            h = host.tolist()
            if h[0] or (h[1] & 1) or any(h[2:]):      # (flags[1] bit 1 only records a re-orthonormalisation: not an error)
                self.check()               # raises (and re-initialises / falls back for the cooperative kernels)
        if self.device.type != "cuda":
            return
        words = [self.flags] + [e[1][0:1] for e in ops._COOP_WS.values()]
        n = sum(w.numel() for w in words)
        if self._poll_host is None or self._poll_host[0].numel() < n:
            self._poll_host = [torch.zeros(max(n, 32), dtype=torch.int32).pin_memory() for _ in range(2)]
        self._poll_host.reverse()          # two pinned buffers, alternating: the one just inspected is the free one
        host = self._poll_host[0][:n]
        o = 0
        for w in words:                    # copy engine, no kernel launch: a few 4-byte D2H copies per poll
            host[o:o + w.numel()].copy_(w, non_blocking=True)
            o += w.numel()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._poll_pending = (host, ev)

    def check(self):
        """raise like trainer.py:240-243 / :341-348 if any step since the last check went bad"""
        self._poll_pending = None
        f = self.flags.tolist()
        self.flags.zero_()
        if f[0]:
            raise ValueError("pred_f2f: non-finite model output")
        if f[1] & 1:
            raise ValueError("Det error: chained rotation with det != 1")
        if self.device.type == "cuda" and not ops.bn_coop_check():
            # (the cooperative BatchNorm launches spin for partner workgroups; a launch that did not find them within its
            #  bound leaves wrong statistics behind.  bn_coop_check has re-initialised the scratch and switched to the
            #  two-launch kernels: the caller restores its last good state and carries on)
            raise RuntimeError("a cooperative BatchNorm launch hit its spin limit: the steps since the last check() are "
                               "invalid; the two-launch BatchNorm kernels are used from here on")
        return f
