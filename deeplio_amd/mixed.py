"""Mixed-precision (bf16 storage) execution of the PointSeg encoders -- BASELINE configs[4]:
"Full DeepLIO bf16 mixed-precision".  No reference counterpart (the reference is fp32 only).

What is bf16: every activation and activation gradient between the stem's max-pool and the global
average pool of an encoder (the Fire blocks, SELayer scaling, max-pools), i.e. the tensors that make
these encoders HBM-bound.  What stays fp32: the master weights and their gradients (flat buffers of
deeplio_amd.optimizer), MFMA accumulation, BatchNorm statistics (fp64 partial sums) and running
statistics, the stem convolution (its input is the fp32 range image), the SELayer's two small fc
layers, and everything behind the encoders' feature vectors: fc1, IMU / odometry RNNs, fusion, heads,
SE(3) chain, loss, optimizer.  Rounding happens once per stored tensor (round to nearest even).

Selected with cfg['lidar-feat-pointseg']['precision'] = 'bf16' (default 'fp32'); the modules of
deeplio_amd.nets dispatch on the dtype of their input, so the same Fire / SELayer / state_dict serve both.

Thin C-ABI wrappers (dlio_*_bf16*) first, torch.autograd.Functions after them; structure mirrors
deeplio_amd.ops / deeplio_amd.functional.
"""
import ctypes as C

import os

import torch
from torch.autograd import Function

from . import ops
from ._lib import check, lib
from .functional import _forked, _sink, _wgrad_stream
from .ops import _ptr, _stream, conv_desc, workspace

BF16 = torch.bfloat16


def _chk16(t):
    return ops._chk(t, BF16)


def _new16(shape, like):
    return torch.empty(shape, dtype=BF16, device=like.device)


# ------------------------------------------------------------------------------------- C-ABI wrappers
def cast(x, to_bf16):
    ops._chk(x, torch.float32 if to_bf16 else BF16)
    y = torch.empty(x.shape, dtype=BF16 if to_bf16 else torch.float32, device=x.device)
    check(lib.dlio_cast_bf16(_ptr(x), _ptr(y), x.numel(), 0 if to_bf16 else 1, _stream()), "cast_bf16")
    return y


def prepped(w, mode):
    """cached bf16 layout of a conv weight (mode 0 forward, 1 data gradient); refreshed for all
    registered weights in one launch per optimizer step (ops._PrepCache family 2)"""
    ops._chk(w)
    return ops._PREP.get(w, mode + 4)


def conv_fwd(x, wt, bias, y, desc, residual=None):
    fn = lib.dlio_conv3x3_bf16_fwd if desc.KH == 3 else lib.dlio_conv1x1_bf16_fwd
    check(fn(_ptr(x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc), _stream()), "conv_bf16_fwd")
    return y


def conv_wgrad(x, dy, dw, desc, accumulate=False):
    nbytes = lib.dlio_conv2d_wgrad_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device)
    check(lib.dlio_conv2d_wgrad_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel(), int(accumulate),
                                     C.byref(desc), _stream()), "conv2d_wgrad_bf16")
    return dw


def _stats_ws(N, C_, HW, device):
    return workspace(lib.dlio_bf16_stats_ws_bytes(N, C_, HW), device, slot=1)


def _partials16(ws, N, C_, HW):
    n = C_ * lib.dlio_bf16_stats_splits(N, C_, HW) * 2
    return ws[:n * 8].view(torch.float64)


# OFF by default: measured slower than the two-launch kernels on this path (1130-1205 vs 1247 frame-pairs/s: at S = 4 a
# channel has 32-128 cooperating workgroups and half the bytes per element, the exchange no longer pays); kept for its test
_BN_COOP16 = [os.environ.get("DLIO_BN_COOP_BF16", "0") != "0"]


def _coop16_ok(N, HW, want_gap):
    """the cooperative one-launch BatchNorm kernels take this geometry (no synchronised statistics: those need the partial
    sums between two launches)"""
    if not _BN_COOP16[0] or not ops._BN_COOP[0] or ops._SYNC_BN[0] is not None:
        return False
    return bool(lib.dlio_bn_coop_gap_ok(N, HW) if want_gap else lib.dlio_bn_coop_ok(N, HW))


def bn_apply(x, x_ctot, x_coff, gamma, beta, eps, momentum, rmean, rvar, y, y_ctot, y_coff, N, C_, HW, post_relu,
             residual=None, r_ctot=0, r_coff=0, gap_out=None, gap_ctot=0, gap_coff=0, eval_prm=None):
    """train: statistics + apply -> prm [3][C] (mean, invstd, scale); eval: apply with eval_prm.  With synchronised
    statistics (GradSync.enable_sync_bn) the per-channel partial sums are all-reduced between the two launches."""
    prm = eval_prm if eval_prm is not None else torch.empty(3, C_, dtype=torch.float32, device=x.device)
    if eval_prm is None and _coop16_ok(N, HW, gap_out is not None):
        # one launch, every operand read once: the workgroups that hold a channel's planes in registers exchange their
        # partial sums (csrc/bn_small.hip, the cooperative kernels over bf16 storage)
        part, sync_ = ops._coop_ws(N, C_, x.device)
        ops._coop_enter(x.device)
        check(lib.dlio_bn_bf16_coop_fwd(_ptr(x), N, x_ctot, x_coff, C_, HW, int(post_relu), _ptr(gamma), _ptr(beta),
                                        float(eps), float(momentum), _ptr(rmean), _ptr(rvar), _ptr(prm[0]), _ptr(prm[1]),
                                        _ptr(prm[2]), _ptr(residual), r_ctot, r_coff, _ptr(y), y_ctot, y_coff, _ptr(gap_out),
                                        gap_ctot, gap_coff, _ptr(part), _ptr(sync_), _stream()), "bn_bf16_coop_fwd")
        ops._coop_exit(x.device)
        return prm
    ws = _stats_ws(N, C_, HW, x.device)

    def call(phase, scale):
        check(lib.dlio_bn_bf16_apply(_ptr(x), N, x_ctot, x_coff, C_, HW, int(post_relu), _ptr(gamma), _ptr(beta),
                                     float(eps), float(momentum), _ptr(rmean), _ptr(rvar), _ptr(prm[0]), _ptr(prm[1]),
                                     _ptr(prm[2]), _ptr(residual), r_ctot, r_coff, _ptr(y), y_ctot, y_coff,
                                     _ptr(gap_out), gap_ctot, gap_coff, int(eval_prm is not None), _ptr(ws), ws.numel(),
                                     phase, float(scale), _stream()), "bn_bf16_apply")
    sync = ops._SYNC_BN[0]
    if sync is None or eval_prm is not None:
        call(0, 1.0)
    else:
        call(1, 1.0)
        sync[0](_partials16(ws, N, C_, HW))
        call(2, sync[1])
    return prm


def bn_bwd(dy, dy_ctot, dy_coff, x, x_ctot, x_coff, prm, beta, dx, dx_ctot, dx_coff, N, C_, HW, post_relu,
           use_batch_stats, dgamma=None, dbeta=None, accumulate=False):
    if use_batch_stats and _coop16_ok(N, HW, False):
        part, sync_ = ops._coop_ws(N, C_, x.device)
        ops._coop_enter(x.device)
        check(lib.dlio_bn_bf16_coop_bwd(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]),
                                        _ptr(prm[2]), _ptr(beta), _ptr(dx), dx_ctot, dx_coff, _ptr(dgamma), _ptr(dbeta),
                                        int(accumulate), N, C_, HW, int(post_relu), _ptr(part), _ptr(sync_), _stream()),
              "bn_bf16_coop_bwd")
        ops._coop_exit(x.device)
        return dx
    ws = _stats_ws(N, C_, HW, x.device)

    def call(phase, scale, local):
        check(lib.dlio_bn_bf16_bwd(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]),
                                   _ptr(prm[2]), _ptr(beta), _ptr(dx), dx_ctot, dx_coff, _ptr(dgamma), _ptr(dbeta),
                                   int(accumulate), N, C_, HW, int(post_relu), int(use_batch_stats), _ptr(ws),
                                   ws.numel(), phase, float(scale), _ptr(local), _stream()), "bn_bf16_bwd")
    sync = ops._SYNC_BN[0]
    if sync is None or not use_batch_stats:
        call(0, 1.0, None)
    else:
        call(1, 1.0, None)
        part = _partials16(ws, N, C_, HW)
        local = part.clone()
        sync[0](part)
        call(2, sync[1], local)
    return dx


def maxpool_fwd(x, k, sh, sw, ph, pw, x_scale=None):
    N, C_, H, W = x.shape
    OH, OW = (H + 2 * ph - k) // sh + 1, (W + 2 * pw - k) // sw + 1
    y = _new16((N, C_, OH, OW), x)
    idx = torch.empty(N, C_, OH, OW, dtype=torch.uint8, device=x.device)
    check(lib.dlio_maxpool_bf16_fwd(_ptr(x), _ptr(x_scale), _ptr(y), _ptr(idx), N, C_, H, W, OH, OW, k, sh, sw, ph,
                                    pw, _stream()), "maxpool_bf16_fwd")
    return y, idx


def maxpool_bwd(dy, idx, in_shape, k, sh, sw, ph, pw, x_scale=None, x_add=None):
    N, C_, H, W = in_shape
    dx = _new16((N, C_, H, W), dy)
    check(lib.dlio_maxpool_bf16_bwd(_ptr(dy), _ptr(idx), _ptr(x_scale), _ptr(x_add), _ptr(dx), N, C_, H, W,
                                    dy.shape[2], dy.shape[3], k, sh, sw, ph, pw, _stream()), "maxpool_bf16_bwd")
    return dx


def maxpool_bwd_dot(dy, idx, x, k, sh, sw, ph, pw):
    N, C_, H, W = x.shape
    ds = torch.empty(N, C_, dtype=torch.float32, device=x.device)
    check(lib.dlio_maxpool_bf16_bwd_dot(_ptr(dy), _ptr(idx), _ptr(x), _ptr(ds), N, C_, H, W, dy.shape[2],
                                        dy.shape[3], k, sh, sw, ph, pw, _stream()), "maxpool_bf16_bwd_dot")
    return ds


def gap_fwd(x, N, ctot, coff, C_, HW):
    out = torch.empty(N, C_, dtype=torch.float32, device=x.device)
    check(lib.dlio_gap_bf16_fwd(_ptr(x), ctot, coff, _ptr(out), N, C_, HW, _stream()), "gap_bf16_fwd")
    return out


def gap_bwd(dout, shape):
    N, C_, H, W = shape
    dx = torch.empty(shape, dtype=BF16, device=dout.device)
    check(lib.dlio_gap_bf16_bwd(_ptr(dout), _ptr(dx), N, C_, H * W, _stream()), "gap_bf16_bwd")
    return dx


# ------------------------------------------------------------------------------------- autograd
class CastFn(Function):
    """fp32 -> bf16 at the entry of the mixed-precision region (gradient: bf16 -> fp32, exact)"""

    @staticmethod
    def forward(ctx, x):
        return cast(x.contiguous(), True)

    @staticmethod
    def backward(ctx, dy):
        return cast(dy.contiguous(), False)


class _CBR16:
    """conv (+bias) -> BatchNorm -> ReLU over channel slices of bf16 buffers (the fp32 path's
    functional._CBR with post_relu, stride 1, 1x1 or 3x3)"""

    @staticmethod
    def forward(x, Cin, H, W, weight, bias, gamma, beta, rmean, rvar, pad, training, momentum, eps, raw, raw_ctot,
                raw_coff, out, out_ctot, out_coff, N, residual=None, r_ctot=0, r_coff=0, gap=None, gap_ctot=0,
                gap_coff=0):
        Cout, _, KH, KW = weight.shape
        d = conv_desc(N, Cin, H, W, Cout, KH, KW, 1, 1, pad, pad, in_ctot=Cin, in_coff=0, out_ctot=raw_ctot,
                      out_coff=raw_coff)
        conv_fwd(x, prepped(weight, 0), bias, raw, d)
        if training:
            d.w16_1 = prepped(weight, 1)       # data-gradient layout, fetched where `weight` is the Parameter
            prm = bn_apply(raw, raw_ctot, raw_coff, gamma, beta, eps, momentum, rmean, rvar, out, out_ctot, out_coff,
                           N, Cout, H * W, True, residual, r_ctot, r_coff, gap, gap_ctot, gap_coff)
        else:
            prm = bn_apply(raw, raw_ctot, raw_coff, gamma, beta, eps, momentum, None, None, out, out_ctot, out_coff,
                           N, Cout, H * W, True, residual, r_ctot, r_coff, gap, gap_ctot, gap_coff,
                           eval_prm=ops.bn_eval_params(rmean, rvar, gamma, eps))
        return d, prm

    @staticmethod
    def backward(dy, dy_ctot, dy_coff, x, d, weight, bias, gamma, prm, beta, raw, training, draw, need_dx, dx=None,
                 dx_ctot=0, dx_coff=0, dx_residual=None, dxr_ctot=0, dxr_coff=0, dx_accumulate=False):
        N, Cout, HW = d.N, d.Cout, d.OH * d.OW
        dgamma, acc_g, ret_g = _sink(gamma, (Cout,), prm)
        dbeta, acc_b, ret_b = _sink(beta, (Cout,), prm)
        if acc_g != acc_b:
            dgamma, dbeta = torch.empty_like(prm[0]), torch.empty_like(prm[0])
            acc_g, ret_g, ret_b = False, dgamma, dbeta
        bn_bwd(dy, dy_ctot, dy_coff, raw, d.out_ctot, d.out_coff, prm, beta, draw, Cout, 0, N, Cout, HW, True,
               training, dgamma, dbeta, accumulate=acc_g)
        ret_bias = None
        if bias is not None:
            dbias, acc, ret_bias = _sink(bias, (Cout,), prm)
            if training:
                if not acc:             # bias in front of a train-mode BN: analytically zero gradient
                    dbias.zero_()
            else:
                ops.chan_sum(cast(draw, False), N, Cout, 0, Cout, HW, out=dbias, accumulate=acc)
        dw, acc_w, ret_w = _sink(weight, weight.shape, prm)
        dd = conv_desc(N, d.Cin, d.H, d.W, Cout, d.KH, d.KW, 1, 1, d.PH, d.PW, in_ctot=d.in_ctot, in_coff=d.in_coff,
                       out_ctot=Cout, out_coff=0)
        ws = _wgrad_stream(dy) if acc_w else None
        if ws is None:
            conv_wgrad(x, draw, dw, dd, accumulate=acc_w)
        else:
            _forked(ws, lambda: conv_wgrad(x, draw, dw, dd, accumulate=True), draw, x)
        if need_dx:
            if dx_accumulate:
                dx_residual, dxr_ctot, dxr_coff = dx, dx_ctot, dx_coff
            w1 = getattr(d, "w16_1", None)
            if w1 is None:
                w1 = prepped(weight, 1)
            g = conv_desc(N, Cout, d.OH, d.OW, d.Cin, d.KH, d.KW, 1, 1, d.KH - 1 - d.PH, d.KW - 1 - d.PW, OH=d.H,
                          OW=d.W, in_ctot=Cout, in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff, res_ctot=dxr_ctot,
                          res_coff=dxr_coff)
            conv_fwd(draw, w1, None, dx, g, residual=dx_residual)
        return ret_w, ret_bias, ret_g, ret_b


class FireFn(Function):
    """Fire block (pointseg_modules.py:116-142) on bf16 tensors: same structure as functional.FireFn
    (squeeze CBR, the two expand convolutions write the halves of the concatenated buffer, 'simple'
    bypass as the residual operand of the BatchNorm apply kernels)."""

    @staticmethod
    def forward(ctx, x, sw, sb, sg, sbe, srm, srv, e1w, e1b, e1g, e1be, e1rm, e1rv, e3w, e3b, e3g, e3be, e3rm, e3rv,
                training, momentum, eps, bypass, want_gap=False):
        ctx.set_materialize_grads(False)     # the plane averages get no gradient: no zero fill in backward
        x = x.contiguous()
        N, Cin, H, W = x.shape
        S_, E1, E3 = sw.shape[0], e1w.shape[0], e3w.shape[0]
        CE = E1 + E3
        raw_s, act_s = _new16((N, S_, H, W), x), _new16((N, S_, H, W), x)
        d_s, prm_s = _CBR16.forward(x, Cin, H, W, sw, sb, sg, sbe, srm, srv, 0, training, momentum, eps, raw_s, S_, 0,
                                    act_s, S_, 0, N)
        raw_e, out = _new16((N, CE, H, W), x), _new16((N, CE, H, W), x)
        res = x if bypass else None
        gap = torch.empty(N, CE, dtype=torch.float32, device=x.device) if want_gap else None
        d_1, prm_1 = _CBR16.forward(act_s, S_, H, W, e1w, e1b, e1g, e1be, e1rm, e1rv, 0, training, momentum, eps,
                                    raw_e, CE, 0, out, CE, 0, N, res, Cin, 0, gap, CE, 0)
        d_3, prm_3 = _CBR16.forward(act_s, S_, H, W, e3w, e3b, e3g, e3be, e3rm, e3rv, 1, training, momentum, eps,
                                    raw_e, CE, E1, out, CE, E1, N, res, Cin, E1, gap, CE, E1)
        ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, prm_1, prm_3, sb, sg,
                              e1b, e1g, e3b, e3g)
        ctx.cfg = (d_s, d_1, d_3, training, bypass)
        if not want_gap:
            return out
        ctx.mark_non_differentiable(gap)
        return out, gap

    @staticmethod
    def backward(ctx, dout, *_unused):
        (x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, prm_1, prm_3, sb, sg, e1b, e1g, e3b,
         e3g) = ctx.saved_tensors
        d_s, d_1, d_3, training, bypass = ctx.cfg
        dout = dout.contiguous()
        N, Cin, H, W = x.shape
        S_, E1, E3 = sw.shape[0], e1w.shape[0], e3w.shape[0]
        CE = E1 + E3
        dact_s = _new16((N, S_, H, W), x)
        draw1 = _new16((N, E1, H, W), x)
        g1 = _CBR16.backward(dout, CE, 0, act_s, d_1, e1w, e1b, e1g, prm_1, e1be, raw_e, training, draw1, True,
                             dact_s, S_, 0)
        del draw1
        draw3 = _new16((N, E3, H, W), x)
        g3 = _CBR16.backward(dout, CE, E1, act_s, d_3, e3w, e3b, e3g, prm_3, e3be, raw_e, training, draw3, True,
                             dact_s, S_, 0, dx_accumulate=True)
        del draw3
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        draw_s = _new16((N, S_, H, W), x)
        gs = _CBR16.backward(dact_s, S_, 0, x, d_s, sw, sb, sg, prm_s, sbe, raw_s, training, draw_s, need_dx, dx,
                             Cin, 0, dout if bypass else None, CE, 0)
        return (dx, gs[0], gs[1], gs[2], gs[3], None, None, g1[0], g1[1], g1[2], g1[3], None, None,
                g3[0], g3[1], g3[2], g3[3], None, None, None, None, None, None, None)


class SEPoolFn(Function):
    """SELayer (pointseg_modules.py:216-221) fused with the max-pool behind it (pointseg_net.py:27-46),
    bf16 tensors; the squeeze / excitation vectors and the two fc layers are fp32"""

    @staticmethod
    def forward(ctx, x, w1, w2, pool, gap=None):
        x = x.contiguous()
        N, C_, H, W = x.shape
        g = gap if gap is not None else gap_fwd(x, N, C_, 0, C_, H * W)
        h = ops.linear_fwd(g, w1, None, ops.ACT_RELU)
        s = ops.linear_fwd(h, w2, None, ops.ACT_SIGMOID)
        k, stride, pad = pool
        y, idx = maxpool_fwd(x, k, stride[0], stride[1], pad[0], pad[1], x_scale=s)
        ctx.save_for_backward(x, w1, w2, g, h, s, idx)
        ctx.pool = pool
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, g, h, s, idx = ctx.saved_tensors
        N, C_, H, W = x.shape
        dy = dy.contiguous()
        k, stride, pad = ctx.pool
        ds = maxpool_bwd_dot(dy, idx, x, k, stride[0], stride[1], pad[0], pad[1])
        dz2 = ops.act_bwd(ds, s, ops.ACT_SIGMOID)
        dw2, acc2, ret2 = _sink(w2, w2.shape, s)
        ops.linear_bwd_weight(dz2, h, N, w2.shape[0], w2.shape[1], dw=dw2, want_bias=False, accumulate=acc2)
        dh = ops.linear_bwd_data(dz2, w2, N)
        dz1 = ops.act_bwd(dh, h, ops.ACT_RELU)
        dw1, acc1, ret1 = _sink(w1, w1.shape, s)
        ops.linear_bwd_weight(dz1, g, N, w1.shape[0], w1.shape[1], dw=dw1, want_bias=False, accumulate=acc1)
        dg = ops.linear_bwd_data(dz1, w1, N)
        dx = maxpool_bwd(dy, idx, tuple(x.shape), k, stride[0], stride[1], pad[0], pad[1], x_scale=s,
                         x_add=ops.ew_scale(dg, 1.0 / (H * W)))
        return dx, ret1, ret2, None, None


class GapFn(Function):
    """adaptive_avg_pool2d(x, (1, 1)).flatten(1): bf16 feature map -> fp32 feature vector (the exit of
    the mixed-precision region)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, C_, H, W = x.shape
        ctx.shape = (N, C_, H, W)
        return gap_fwd(x, N, C_, 0, C_, H * W)

    @staticmethod
    def backward(ctx, dy):
        return gap_bwd(dy.contiguous(), ctx.shape)
