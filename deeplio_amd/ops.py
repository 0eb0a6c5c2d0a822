"""Thin tensor-level wrappers over the C-ABI (no autograd, no math): they extract device
pointers from torch storage, pass the current HIP stream, own the scratch workspace and
translate status codes.  torch is used here only for device memory and streams.
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import ConvDesc, check, lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def raw_stream():
    """handle (int) of torch's current HIP stream on the current device.  torch.cuda.current_stream()
    builds a Python Stream object through four layers of device-index helpers (~9 us; at ~1100 launches
    per step that was 5 ms of host time on a step whose host side is nearly critical) -- this is the
    C-level getter the same call ends in"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def streams_share_queue(a, b):
    """True when torch streams a and b are served by one hardware queue (dlio_streams_share_queue; synchronises both)"""
    out = C.c_int(0)
    check(lib.dlio_streams_share_queue(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(out)), "streams_share_queue")
    return bool(out.value)


def _stream():
    return C.c_void_p(raw_stream())


def _chk(t, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("deeplio_amd ops need HIP device tensors (got %s); there is no CPU "
                           "fallback" % t.device)
    if t.dtype != dtype:
        raise ValueError("expected %s tensor, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return t


_WS = {}


def workspace(nbytes, device, slot=0):
    """Caller-owned scratch handed to the library (grown geometrically, reused).  One slot
    per (device, stream, slot) so concurrent streams never share scratch."""
    key = (device.index if device.index is not None else torch._C._cuda_getDevice(), raw_stream(), slot)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        size = max(int(nbytes * 1.25), 1 << 20)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ----------------------------------------------------------------------------- conv
def conv_desc(N, Cin, H, W, Cout, KH, KW, SH, SW, PH, PW, OH=None, OW=None, in_ctot=None,
              in_coff=0, out_ctot=None, out_coff=0, res_ctot=0, res_coff=0, in_relu=0):
    if OH is None:
        OH = (H + 2 * PH - KH) // SH + 1
    if OW is None:
        OW = (W + 2 * PW - KW) // SW + 1
    d = ConvDesc()
    d.N, d.Cin, d.H, d.W = N, Cin, H, W
    d.in_ctot, d.in_coff = (Cin if in_ctot is None else in_ctot), in_coff
    d.Cout, d.OH, d.OW = Cout, OH, OW
    d.out_ctot, d.out_coff = (Cout if out_ctot is None else out_ctot), out_coff
    d.KH, d.KW, d.SH, d.SW, d.PH, d.PW = KH, KW, SH, SW, PH, PW
    d.res_ctot, d.res_coff, d.in_relu = res_ctot, res_coff, in_relu
    return d


def conv2d_prep_weight(w, mode, out=None):
    _chk(w)
    Cout, Cin, KH, KW = w.shape
    if out is None:
        out = torch.empty(lib.dlio_conv2d_prep_weight_floats(Cout, Cin, KH, KW, mode), dtype=torch.float32,
                          device=w.device)
    check(lib.dlio_conv2d_prep_weight(_ptr(w), _ptr(out), Cout, Cin, KH, KW, mode, _stream()),
          "conv2d_prep_weight")
    return out


# ---- prepped-weight cache: all conv weights of the model are re-laid-out in ONE launch ----------
# Weights change once per optimizer step; forward needs layout 0 and backward layout 1 of every
# conv weight.  Instead of 2 small launches per conv and step (146 for PointSeg), every weight that
# passes through `conv2d_prepped` is registered, and the first request of a new "weights epoch"
# refreshes ALL registered buffers with dlio_conv2d_prep_weights_batched.  Staleness is detected by
# (epoch, tensor._version): the epoch is bumped by our optimizer kernels (`weights_changed`), the
# version counter by any in-place torch op (load_state_dict, copy_, broadcast).
import ctypes as _ct
import weakref as _weakref


class _PrepItemC(_ct.Structure):
    _fields_ = [("w", _ct.c_void_p), ("wt", _ct.c_void_p), ("Cout", _ct.c_int32), ("Cin", _ct.c_int32),
                ("taps", _ct.c_int32), ("mode", _ct.c_int32), ("start", _ct.c_int64)]


class _PrepCache:
    """mode 0 / 1: fp32-MFMA layouts (forward / data gradient) of any conv weight; mode 2 / 3: the
    split-bf16 layouts (conv_bx3.hip) of 3x3 / 1x1 weights, forward / data gradient; mode 4 / 5: the
    native bf16 layouts of the mixed-precision path (conv_bf16.hip); mode 6 / 7: two fp16 pieces of w 2^k + the scales
    (fire_expand.hip)"""

    def __init__(self):
        self.epoch = 0
        self.entries = {}        # (device index, data_ptr, shape, mode) -> dict
        self.tables = {}         # (device index, family) -> dict(items_dev, n, total, keys)
        self.waited = {}         # (stream handle, id of a refresh event) it has waited for (cleared by every refresh)
        self.batched = True
        self.side = None         # callable(current stream) -> a stream for the long two-piece / bf16 family launches, or None

    @staticmethod
    def _floats(w, mode):
        Cout, Cin, KH, KW = w.shape
        if mode >= 6:
            return lib.dlio_conv_h2_prep_floats(Cout, Cin, KH * KW, mode - 6)
        if mode >= 4:
            return (lib.dlio_conv_bf16_prep_elems(Cout, Cin, KH * KW, mode - 4) + 1) // 2     # bf16 in fp32 storage
        if mode >= 2:
            return lib.dlio_conv_bx3_prep_floats(Cout, Cin, KH * KW, mode - 2)
        return lib.dlio_conv2d_prep_weight_floats(Cout, Cin, KH, KW, mode)

    @staticmethod
    def _units(w, mode):
        """index-space size of one item in its family's batched kernel"""
        Cout, Cin, KH, KW = w.shape
        if mode >= 2:
            K, Nn = (Cin, Cout) if mode in (2, 4, 6) else (Cout, Cin)
            return KH * KW * ((K + 15) // 16) * Nn * 16
        return lib.dlio_conv2d_prep_weight_floats(Cout, Cin, KH, KW, mode)

    def get(self, w, mode):
        dev = w.device.index if w.device.index is not None else torch.cuda.current_device()
        key = (dev, w.data_ptr(), tuple(w.shape), mode)
        e = self.entries.get(key)
        if e is None or e["ref"]() is None:
            e = dict(ref=_weakref.ref(w), epoch=-1, version=-1, stream=None, event=None,
                     out=torch.empty(self._floats(w, mode), dtype=torch.float32, device=w.device))
            self.entries[key] = e
            self.tables.pop((dev, mode >> 1), None)
        cur_raw = raw_stream()
        if e["epoch"] != self.epoch or e["version"] != w._version:
            cur = torch.cuda.current_stream()
            if self.batched and e["epoch"] >= 0:          # a known weight went stale: refresh them all
                self._refresh_all(dev, cur)
            if e["epoch"] != self.epoch or e["version"] != w._version:     # newly registered / batching off
                Cout, Cin, KH, KW = w.shape
                if mode >= 6:
                    check(lib.dlio_conv_h2_prep(_ptr(w), _ptr(e["out"]), Cout, Cin, KH * KW, mode - 6, _stream()),
                          "conv_h2_prep")
                elif mode >= 4:
                    check(lib.dlio_conv_bf16_prep(_ptr(w), _ptr(e["out"]), Cout, Cin, KH * KW, mode - 4, _stream()),
                          "conv_bf16_prep")
                elif mode >= 2:
                    check(lib.dlio_conv_bx3_prep(_ptr(w), _ptr(e["out"]), Cout, Cin, KH * KW, mode - 2, _stream()),
                          "conv_bx3_prep")
                else:
                    check(lib.dlio_conv2d_prep_weight(_ptr(w), _ptr(e["out"]), Cout, Cin, KH, KW, mode, _stream()),
                          "conv2d_prep_weight")
                e.update(epoch=self.epoch, version=w._version, stream=cur_raw, event=None)
        if e["stream"] != cur_raw and e["event"] is not None:
            # prepped on another stream in this epoch: one wait per (stream, refresh event)
            wk = (cur_raw, id(e["event"]))
            if wk not in self.waited:
                torch.cuda.current_stream().wait_event(e["event"])
                self.waited[wk] = True
        return e["out"]

    def _table(self, dev, family):
        t = self.tables.get((dev, family))
        if t is None:
            live = [(k, e) for k, e in self.entries.items() if k[0] == dev and (k[3] >> 1) == family
                    and e["ref"]() is not None and e["epoch"] >= 0]       # only weights that were used before
            if not live:
                return None
            arr = (_PrepItemC * len(live))()
            start = 0
            for i, (k, e) in enumerate(live):
                w = e["ref"]()
                Cout, Cin, KH, KW = w.shape
                arr[i] = _PrepItemC(w.data_ptr(), e["out"].data_ptr(), Cout, Cin, KH * KW, k[3] & 1, start)
                start += self._units(w, k[3])
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            t = dict(items=raw.to(torch.device("cuda", dev)), n=len(live), total=start, keys=[k for k, _ in live])
            self.tables[(dev, family)] = t
        return t

    def _refresh_all(self, dev, cur):
        """rebuild every registered layout (the optimizer changed the weights): the fp32-MFMA and three-piece families -- what
        the stems and the first squeeze layers read -- on the calling stream; the two-piece / bf16 families (the long launches:
        a magnitude pass + the split of every Fire and plain 3x3 weight, ~230 us at the headline shape) on `self.side(cur)`,
        a stream that is idle at the head of a step (the caller's weight-gradient companion), so that the stems do not queue
        behind them; every entry carries its family's event, a reader on another stream waits for it once"""
        for k in [k for k, e in self.entries.items() if e["ref"]() is None]:
            del self.entries[k]
            self.tables.pop((k[0], k[3] >> 1), None)
        self.waited.clear()
        side = self.side(cur) if self.side is not None else None
        groups = ((cur, ((0, lib.dlio_conv2d_prep_weights_batched, "conv2d_prep_weights_batched"),
                         (1, lib.dlio_conv3x3_bx3_prep_batched, "conv3x3_bx3_prep_batched"))),
                  (side if side is not None else cur,
                   ((2, lib.dlio_conv_bf16_prep_batched, "conv_bf16_prep_batched"),
                    (3, lib.dlio_conv_h2_prep_batched, "conv_h2_prep_batched"))))
        for st, fams in groups:
            done = []
            other = st is not cur and st.cuda_stream != cur.cuda_stream
            if other:
                st.wait_stream(cur)                        # the weights are final on the calling stream
                prev = torch._C._cuda_getCurrentStream(st.device_index)
                torch._C._cuda_setStream(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)
            try:
                for family, fn, name in fams:
                    t = self._table(dev, family)
                    if t is None:
                        continue
                    check(fn(_ptr(t["items"]), t["n"], t["total"], _stream()), name)
                    done += t["keys"]
            finally:
                if other:
                    torch._C._cuda_setStream(stream_id=prev[0], device_index=prev[1], device_type=prev[2])
            if not done:
                continue
            ev = torch.cuda.Event()
            ev.record(st)
            for k in done:
                e = self.entries.get(k)
                w = e["ref"]() if e is not None else None
                if w is not None:
                    e.update(epoch=self.epoch, version=w._version, stream=st.cuda_stream, event=ev)


_PREP = _PrepCache()


def weights_changed():
    """call after parameters were modified through raw pointers (our optimizer kernels)"""
    _PREP.epoch += 1


def conv2d_prepped(w, mode):
    """cached dlio_conv2d_prep_weight(w, mode); see _PrepCache"""
    _chk(w)
    return _PREP.get(w, mode)


def conv2d_fwd(x, wt, bias, y, desc, in_aff=None, residual=None):
    """in_aff = (mean, scale, shift) per input channel or None."""
    m = s = b = None
    if in_aff is not None:
        m, s, b = in_aff
    check(lib.dlio_conv2d_fwd(_ptr(x), _ptr(wt), _ptr(bias), _ptr(m), _ptr(s), _ptr(b),
                              _ptr(residual), _ptr(y), C.byref(desc), _stream()), "conv2d_fwd")
    return y


def conv3x3_bx3_prep(w, mode, out=None):
    """split-bf16 weight layout of a [Cout, Cin, 3, 3] tensor (mode 0 forward, 1 data gradient)"""
    _chk(w)
    Cout, Cin, KH, KW = w.shape
    if (KH, KW) != (3, 3):
        raise ValueError("conv3x3_bx3 needs a 3x3 kernel")
    if out is None:
        out = torch.empty(lib.dlio_conv3x3_bx3_prep_floats(Cout, Cin, mode), dtype=torch.float32, device=w.device)
    check(lib.dlio_conv3x3_bx3_prep(_ptr(w), _ptr(out), Cout, Cin, mode, _stream()), "conv3x3_bx3_prep")
    return out


def conv_bx3_prepped(w, mode):
    """cached split-bf16 layout of a 3x3 or 1x1 weight (mode 0 forward, 1 data gradient); see
    _PrepCache.  Call with the long-lived Parameter (forward); backward receives the layouts on the
    ConvDesc."""
    _chk(w)
    if tuple(w.shape[2:]) not in ((3, 3), (1, 1), (3, 5)):     # (3x3 also covers the stride-2 forward: same tap layout)
        raise ValueError("split-bf16 kernels exist for 3x3, 1x1 and (forward, stride (1, 2)) 3x5 weights")
    return _PREP.get(w, mode + 2)


conv3x3_bx3_prepped = conv_bx3_prepped


def conv_h2_prepped(w, mode=0):
    """cached two-piece fp16 layout of a 3x3 or 1x1 weight (dlio_conv_h2_prep; the fused Fire forward's planes_fmt 1)"""
    _chk(w)
    return _PREP.get(w, mode + 6)


def conv1x1_bx3_prep(w, mode, out=None):
    _chk(w)
    Cout, Cin, KH, KW = w.shape
    if (KH, KW) != (1, 1):
        raise ValueError("conv1x1_bx3 needs a 1x1 kernel")
    if out is None:
        out = torch.empty(lib.dlio_conv_bx3_prep_floats(Cout, Cin, 1, mode), dtype=torch.float32, device=w.device)
    check(lib.dlio_conv_bx3_prep(_ptr(w), _ptr(out), Cout, Cin, 1, mode, _stream()), "conv_bx3_prep")
    return out


def conv1x1_bx3_fwd(x, wt, bias, y, desc, residual=None, in_aff=None):
    """in_aff = (mean, scale, shift) per input channel or None (apply-on-load, see conv2d_fwd)"""
    m = s = b = None
    if in_aff is not None:
        m, s, b = in_aff
    nbytes = lib.dlio_conv1x1_bx3_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device, slot=4) if nbytes else None       # K split over workgroups (narrowing small layers)
    check(lib.dlio_conv1x1_bx3_fwd_ws(_ptr(x), _ptr(wt), _ptr(bias), _ptr(m), _ptr(s), _ptr(b), _ptr(residual), _ptr(y),
                                      _ptr(ws), ws.numel() if ws is not None else 0, C.byref(desc), _stream()),
          "conv1x1_bx3_fwd")
    return y


def conv1x1_h2_fwd(x, amax_x, wt, bias, y, desc, residual=None):
    """the 1x1 convolution on the two-piece fp16 split: amax_x = one-float tensor with the largest |x| (or a bound), wt =
    conv_h2_prepped(w, mode)"""
    nbytes = lib.dlio_conv1x1_bx3_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device, slot=4) if nbytes else None
    check(lib.dlio_conv1x1_h2_fwd(_ptr(x), _ptr(amax_x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), _ptr(ws),
                                  ws.numel() if ws is not None else 0, C.byref(desc), _stream()), "conv1x1_h2_fwd")
    return y


def conv3x3_bx3_fwd(x, wt, bias, y, desc, residual=None):
    nbytes = lib.dlio_conv3x3_bx3_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device, slot=4) if nbytes else None       # K split over workgroups (small feature maps)
    check(lib.dlio_conv3x3_bx3_fwd_ws(_ptr(x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), _ptr(ws),
                                      ws.numel() if ws is not None else 0, C.byref(desc), _stream()), "conv3x3_bx3_fwd")
    return y


def fire_expand_dgrad(dy3, wt3, dy1, wt1, dx, desc, residual=None):
    """dx = conv3x3(dy3, wt3) + conv1x1(dy1, wt1) (+ residual): both expand data gradients of a Fire block in one launch
    (wt3 / wt1 = conv_bx3_prepped(w, 1) of the two layers; dy1 contiguous)"""
    nbytes = lib.dlio_conv3x3_bx3_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, dy3.device, slot=4) if nbytes else None
    check(lib.dlio_fire_expand_dgrad(_ptr(dy3), _ptr(wt3), _ptr(dy1), _ptr(wt1), dy1.shape[1], _ptr(residual), _ptr(dx),
                                     _ptr(ws), ws.numel() if ws is not None else 0, C.byref(desc), _stream()),
          "fire_expand_dgrad")
    return dx


def conv3x5s2_bx3_fwd(x, wt, bias, y, desc, residual=None):
    """the PointSeg stem (3x5 taps, stride (1, 2)) on the split-bf16 kernel; wt = conv_bx3_prepped(w, 0)"""
    check(lib.dlio_conv3x5s2_bx3_fwd(_ptr(x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc), _stream()),
          "conv3x5s2_bx3_fwd")
    return y


def phase_interleave2d(phases, SH, SW, dx, dx_ctot, dx_coff, N, C_, H, W, residual=None, r_ctot=0, r_coff=0):
    """phases: SH*SW tensors [N, C, ceil((H-a)/SH), ceil((W-b)/SW)] (index a*SW+b, None = all zero)"""
    arr = (C.c_void_p * 4)(*[(None if t is None else t.data_ptr()) for t in list(phases) + [None] * (4 - len(phases))])
    check(lib.dlio_phase_interleave2d(arr, SH, SW, _ptr(residual), r_ctot, r_coff, _ptr(dx), dx_ctot, dx_coff,
                                      N, C_, H, W, _stream()), "phase_interleave2d")
    return dx


# data-gradient layouts of the tap subsets of strided convolutions (one per phase).  Cached per
# weight OBJECT (weakref: a data_ptr can be reused by another tensor) and refreshed when the weight
# changes; only the long-lived Parameter seen in forward can be cached -- backward sees unpacked
# copies, so the layouts are fetched in forward and travel on the ConvDesc (like `wt2`).
_PHASE_W = {}


def conv2d_prepped_phase(w, SH, SW, rh, rw, cache=True):
    """data-gradient (mode 1) layout of w[:, :, rh::SH, rw::SW]"""
    Cout, Cin, KH, KW = w.shape
    Mh, Mw = len(range(rh, KH, SH)), len(range(rw, KW, SW))
    nfl = lib.dlio_conv2d_prep_weight_floats(Cout, Cin, Mh, Mw, 1)
    e = None
    if cache:
        key = (w.data_ptr(), tuple(w.shape), SH, SW, rh, rw)
        e = _PHASE_W.get(key)
        if e is None or e["ref"]() is not w:
            e = _PHASE_W[key] = dict(ref=_weakref.ref(w), epoch=-1, version=-1,
                                     out=torch.empty(nfl, dtype=torch.float32, device=w.device))
            for k in [k for k, v in _PHASE_W.items() if v["ref"]() is None]:
                del _PHASE_W[k]
        if e["epoch"] == _PREP.epoch and e["version"] == w._version:
            return e["out"]
        out = e["out"]
    else:
        out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    sub = w.detach()[:, :, rh::SH, rw::SW].contiguous()
    check(lib.dlio_conv2d_prep_weight(_ptr(sub), _ptr(out), Cout, Cin, Mh, Mw, 1, _stream()), "conv2d_prep_weight")
    sub.record_stream(torch.cuda.current_stream())
    if e is not None:
        e.update(epoch=_PREP.epoch, version=w._version)
    return out


def conv_bx3_prepped_phase(w, SH, SW, rh, rw, cache=True):
    """split-bf16 data-gradient (mode 1) layout of w[:, :, rh::SH, rw::SW] (dlio_conv_bx3_fwd_taps)"""
    Cout, Cin, KH, KW = w.shape
    Mh, Mw = len(range(rh, KH, SH)), len(range(rw, KW, SW))
    nfl = lib.dlio_conv_bx3_prep_floats(Cout, Cin, Mh * Mw, 1)
    e = None
    if cache:
        key = ("bx3", w.data_ptr(), tuple(w.shape), SH, SW, rh, rw)
        e = _PHASE_W.get(key)
        if e is None or e["ref"]() is not w:
            e = _PHASE_W[key] = dict(ref=_weakref.ref(w), epoch=-1, version=-1,
                                     out=torch.empty(nfl, dtype=torch.float32, device=w.device))
            for k in [k for k, v in _PHASE_W.items() if v["ref"]() is None]:
                del _PHASE_W[k]
        if e["epoch"] == _PREP.epoch and e["version"] == w._version:
            return e["out"]
        out = e["out"]
    else:
        out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    sub = w.detach()[:, :, rh::SH, rw::SW].contiguous()
    check(lib.dlio_conv_bx3_prep(_ptr(sub), _ptr(out), Cout, Cin, Mh * Mw, 1, _stream()), "conv_bx3_prep")
    sub.record_stream(torch.cuda.current_stream())
    if e is not None:
        e.update(epoch=_PREP.epoch, version=w._version)
    return out


def conv_h2_prepped_phase(w, SH, SW, rh, rw, cache=True):
    """two-piece fp16 data-gradient (mode 1) layout of w[:, :, rh::SH, rw::SW] (dlio_conv_h2_fwd_taps)"""
    Cout, Cin, KH, KW = w.shape
    Mh, Mw = len(range(rh, KH, SH)), len(range(rw, KW, SW))
    nfl = lib.dlio_conv_h2_prep_floats(Cout, Cin, Mh * Mw, 1)
    e = None
    if cache:
        key = ("h2", w.data_ptr(), tuple(w.shape), SH, SW, rh, rw)
        e = _PHASE_W.get(key)
        if e is None or e["ref"]() is not w:
            e = _PHASE_W[key] = dict(ref=_weakref.ref(w), epoch=-1, version=-1,
                                     out=torch.empty(nfl, dtype=torch.float32, device=w.device))
            for k in [k for k, v in _PHASE_W.items() if v["ref"]() is None]:
                del _PHASE_W[k]
        if e["epoch"] == _PREP.epoch and e["version"] == w._version:
            return e["out"]
        out = e["out"]
    else:
        out = torch.empty(nfl, dtype=torch.float32, device=w.device)
    sub = w.detach()[:, :, rh::SH, rw::SW].contiguous()
    check(lib.dlio_conv_h2_prep(_ptr(sub), _ptr(out), Cout, Cin, Mh * Mw, 1, _stream()), "conv_h2_prep")
    sub.record_stream(torch.cuda.current_stream())
    if e is not None:
        e.update(epoch=_PREP.epoch, version=w._version)
    return out


def conv_h2_strided_fwd(x, amax_x, wt, bias, y, desc, residual=None):
    """3x5 stride (1, 2) / 3x3 stride (2, 2) forward on the two-piece fp16 split (more than 32 output channels)"""
    check(lib.dlio_conv_h2_fwd_strided(_ptr(x), _ptr(amax_x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc),
                                       _stream()), "conv_h2_fwd_strided")
    return y


def conv_h2_taps_fwd(x, amax_x, wt, bias, y, desc, residual=None):
    """conv_bx3_taps_fwd on the two-piece fp16 split"""
    check(lib.dlio_conv_h2_fwd_taps(_ptr(x), _ptr(amax_x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc),
                                    _stream()), "conv_h2_fwd_taps")
    return y


def conv_bx3_taps_fwd(x, wt, bias, y, desc, residual=None):
    """stride-1 convolution with a small tap window and an explicit output extent on the split-bf16 kernel"""
    check(lib.dlio_conv_bx3_fwd_taps(_ptr(x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc), _stream()),
          "conv_bx3_fwd_taps")
    return y


def zero_upsample2d(src, HU, WU, SH, SW):
    """[N,C,OH,OW] -> [N,C,HU,WU] with the stride's zeros inserted"""
    _chk(src)
    N, C_, OH, OW = src.shape
    dst = torch.empty(N, C_, HU, WU, dtype=torch.float32, device=src.device)
    check(lib.dlio_zero_upsample2d(_ptr(src), _ptr(dst), N * C_, OH, OW, HU, WU, SH, SW, _stream()),
          "zero_upsample2d")
    return dst


def conv2d_dgrad_strided(dy, w, dx, desc):
    check(lib.dlio_conv2d_dgrad_strided(_ptr(dy), _ptr(w), _ptr(dx), C.byref(desc), _stream()),
          "conv2d_dgrad_strided")
    return dx


def conv2d_wgrad(x, dy, dw, desc, in_aff=None, accumulate=False):
    m = s = b = None
    if in_aff is not None:
        m, s, b = in_aff
    nbytes = lib.dlio_conv2d_wgrad_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device)
    check(lib.dlio_conv2d_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(m), _ptr(s), _ptr(b), _ptr(ws),
                                ws.numel(), int(accumulate), C.byref(desc), _stream()), "conv2d_wgrad")
    return dw


def conv3x3_wgrad_h2_ok(desc):
    return bool(lib.dlio_conv3x3_wgrad_h2_ok(C.byref(desc)))


def conv3x3_wgrad_h2(x, amax_x, dy, amax_dy, dw, desc, accumulate=False):
    """3x3 stride-1 weight gradient on the two-piece fp16 split; amax_x / amax_dy: one-float device tensors with the
    operands' largest magnitudes or bounds on them (bn_split16 bound_out, bn_coop_bwd amax_out)"""
    nbytes = lib.dlio_conv2d_wgrad_ws_bytes(C.byref(desc))
    ws = workspace(nbytes, x.device)
    check(lib.dlio_conv3x3_wgrad_h2(_ptr(x), _ptr(amax_x), _ptr(dy), _ptr(amax_dy), _ptr(dw), _ptr(ws), ws.numel(),
                                    int(accumulate), C.byref(desc), _stream()), "conv3x3_wgrad_h2")
    return dw


# ----------------------------------------------------------------------------- batch norm
def _stats_ws(N, C_, HW, device):
    nbytes = lib.dlio_chan_stats_ws_bytes(N, C_, HW)
    return workspace(nbytes, device, slot=1)


def chan_stats(x, N, ctot, coff, C_, HW, pre_relu=False):
    s = torch.empty(2, C_, dtype=torch.float64, device=x.device)
    ws = _stats_ws(N, C_, HW, x.device)
    check(lib.dlio_chan_stats(_ptr(x), N, ctot, coff, C_, HW, int(pre_relu), _ptr(s[0]), _ptr(s[1]),
                              _ptr(ws), ws.numel(), _stream()), "chan_stats")
    return s


def bn_finalize(stats, count, gamma, eps, momentum, running_mean, running_var):
    """-> params [3][C]: mean, invstd, scale"""
    C_ = stats.shape[1]
    prm = torch.empty(3, C_, dtype=torch.float32, device=stats.device)
    check(lib.dlio_bn_finalize(_ptr(stats[0]), _ptr(stats[1]), C_, float(count), _ptr(gamma),
                               float(eps), float(momentum), _ptr(running_mean), _ptr(running_var),
                               _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _stream()), "bn_finalize")
    return prm


def bn_train_stats(x, N, ctot, coff, C_, HW, pre_relu, gamma, eps, momentum, running_mean, running_var, prm=None,
                   beta=None, shift_out=None):
    """batch statistics + finalize (+ running-stat update) -> params [3][C]: mean, invstd, scale (prm: three
    caller-provided [C] tensors to write them into)"""
    if prm is None:
        prm = torch.empty(3, C_, dtype=torch.float32, device=x.device)
    ws = _stats_ws(N, C_, HW, x.device)

    def call(phase, scale):
        check(lib.dlio_bn_train_stats(_ptr(x), N, ctot, coff, C_, HW, int(pre_relu), _ptr(gamma), float(eps),
                                      float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(prm[0]),
                                      _ptr(prm[1]), _ptr(prm[2]), _ptr(ws), ws.numel(), _ptr(beta), _ptr(shift_out),
                                      phase, float(scale), _stream()), "bn_train_stats")
    sync = _SYNC_BN[0]
    if sync is None:
        call(0, 1.0)
    else:                       # synchronised statistics: partials all-reduced between the two launches
        call(1, 1.0)
        sync[0](_partials_view(ws, N, C_, HW))
        call(2, sync[1])
    return prm


# SyncBN: (all_reduce callable, world) or None.  The callable sums a float64 tensor over the
# data-parallel replicas in place (deeplio_amd.dist.GradSync.enable_sync_bn).
_SYNC_BN = [None]


def set_sync_bn(all_reduce, world):
    _SYNC_BN[0] = (all_reduce, int(world)) if all_reduce is not None and world > 1 else None


def _partials_view(ws, N, C_, HW):
    n = C_ * lib.dlio_chan_stats_splits(N, C_, HW) * 2
    return ws[:n * 8].view(torch.float64)


def bn_train_apply(x, x_ctot, x_coff, gamma, beta, eps, momentum, running_mean, running_var, y, y_ctot,
                   y_coff, N, C_, HW, pre_relu, post_relu, residual=None, r_ctot=0, r_coff=0,
                   gap_out=None, gap_ctot=0, gap_coff=0, r_aff=None, amax_out=None):
    """train-mode BN forward (statistics + apply) in two launches -> prm [3][C]; with SyncBN the
    partial sums are all-reduced between the two launches.  amax_out: a zeroed one-float tensor that receives max |y|"""
    prm = torch.empty(3, C_, dtype=torch.float32, device=x.device)
    ws = _stats_ws(N, C_, HW, x.device)

    def call(phase, scale):
        check(lib.dlio_bn_train_apply(_ptr(x), N, x_ctot, x_coff, C_, HW, int(pre_relu), int(post_relu),
                                      _ptr(gamma), _ptr(beta), float(eps), float(momentum), _ptr(running_mean),
                                      _ptr(running_var), _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]),
                                      _ptr(residual), r_ctot, r_coff, _ptr(y), y_ctot, y_coff, _ptr(gap_out),
                                      gap_ctot, gap_coff, _ptr(ws), ws.numel(), phase, float(scale),
                                      _ptr(r_aff[0]) if r_aff is not None else None,
                                      _ptr(r_aff[1]) if r_aff is not None else None,
                                      _ptr(r_aff[2]) if r_aff is not None else None, _ptr(amax_out) if phase != 1 else None,
                                      _stream()),
              "bn_train_apply")
    sync = _SYNC_BN[0]
    if sync is None:
        call(0, 1.0)
    else:
        call(1, 1.0)
        sync[0](_partials_view(ws, N, C_, HW))
        call(2, sync[1])
    return prm


def bn_small_ok(N, HW):
    """the one-launch BatchNorm kernels of csrc/bn_small.hip take this geometry (and SyncBN is off: synchronised
    statistics need the partial sums between two launches)"""
    return _SYNC_BN[0] is None and bool(lib.dlio_bn_small_ok(N, HW))


def bn_small_fwd(x, x_ctot, x_coff, N, C_, C1, HW, set1, set2, eps, momentum, prm, y, y_ctot, y_coff, post_relu=True,
                 shift_out=None, residual=None, r_ctot=0, r_coff=0, r_aff=None, gap_out=None, gap_ctot=0, gap_coff=0,
                 amax_out=None):
    """train-mode BatchNorm (+ ReLU, + residual) of a small feature map in one launch; set1 / set2 = (gamma, beta,
    running_mean, running_var) of the channels [0, C1) / [C1, C_) (set2 None: one layer); prm = (mean, invstd, scale)
    rows of C_ floats; y None: statistics only"""
    g2 = set2 if set2 is not None else (None, None, None, None)
    check(lib.dlio_bn_small_fwd(_ptr(x), N, x_ctot, x_coff, C_, C1, HW, int(post_relu), _ptr(set1[0]), _ptr(set1[1]),
                                _ptr(set1[2]), _ptr(set1[3]), _ptr(g2[0]), _ptr(g2[1]), _ptr(g2[2]), _ptr(g2[3]), float(eps),
                                float(momentum), _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(shift_out), _ptr(residual),
                                r_ctot, r_coff, _ptr(r_aff[0]) if r_aff is not None else None,
                                _ptr(r_aff[1]) if r_aff is not None else None, _ptr(r_aff[2]) if r_aff is not None else None,
                                _ptr(y), y_ctot, y_coff, _ptr(gap_out), gap_ctot, gap_coff, _ptr(amax_out), _stream()),
          "bn_small_fwd")
    return prm


def bn_small_bwd(dy, dy_ctot, dy_coff, x, x_ctot, x_coff, prm, beta1, beta2, dx1, dx2, dg1, db1, dg2, db2, accumulate, N,
                 C_, C1, HW, post_relu=True, amax_out=None):
    """amax_out: a zeroed one-float tensor (amax_slot) that receives max |dx| (for the two-piece split kernels)"""
    check(lib.dlio_bn_small_bwd(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]),
                                _ptr(beta1), _ptr(beta2), _ptr(dx1), _ptr(dx2), _ptr(dg1), _ptr(db1), _ptr(dg2), _ptr(db2),
                                int(accumulate), N, C_, C1, HW, int(post_relu), _ptr(amax_out), _stream()), "bn_small_bwd")


def bn_aff_apply(x, x_ctot, x_coff, N, C_, HW, aff, y, y_ctot, y_coff, residual=None, r_ctot=0, r_coff=0, r_aff=None,
                 gap_out=None, gap_ctot=0, gap_coff=0):
    """streaming BatchNorm + ReLU (+ residual, itself apply-on-load with r_aff) from known statistics: aff = (mean, scale,
    shift) rows over the C_ channels (csrc/bn_stream.hip)"""
    ra = r_aff if r_aff is not None else (None, None, None)
    check(lib.dlio_bn_aff_apply(_ptr(x), N, x_ctot, x_coff, C_, HW, _ptr(aff[0]), _ptr(aff[1]), _ptr(aff[2]), _ptr(residual),
                                r_ctot, r_coff, _ptr(ra[0]), _ptr(ra[1]), _ptr(ra[2]), _ptr(y), y_ctot, y_coff, _ptr(gap_out),
                                gap_ctot, gap_coff, _stream()), "bn_aff_apply")
    return y


def bn_aff_pool_ok(H, W, SH):
    return bool(lib.dlio_bn_aff_pool_ok(H, W, SH))


def bn_aff_pool_fwd(x, x_ctot, x_coff, N, C_, H, W, SH, aff, residual=None, r_ctot=0, r_coff=0, r_aff=None, want_gap=True):
    """BatchNorm + ReLU (+ residual) + MaxPool2d(3, (SH, 2), 1) without the full-resolution output -> (pooled maximum,
    arg-max codes, plane averages of the un-pooled output); the SELayer's scale is applied to the POOLED tensor afterwards"""
    OH, OW = (H + 2 - 3) // SH + 1, W // 2
    yp = torch.empty(N, C_, OH, OW, dtype=torch.float32, device=x.device)
    idx = torch.empty(N, C_, OH, OW, dtype=torch.uint8, device=x.device)
    gap = torch.empty(N, C_, dtype=torch.float32, device=x.device) if want_gap else None
    ra = r_aff if r_aff is not None else (None, None, None)
    check(lib.dlio_bn_aff_pool_fwd(_ptr(x), N, x_ctot, x_coff, C_, H, W, SH, _ptr(aff[0]), _ptr(aff[1]), _ptr(aff[2]),
                                   _ptr(residual), r_ctot, r_coff, _ptr(ra[0]), _ptr(ra[1]), _ptr(ra[2]), _ptr(yp), _ptr(idx),
                                   _ptr(gap), C_, 0, _stream()), "bn_aff_pool_fwd")
    return yp, idx, gap


_COOP_WS = {}
_BN_COOP = [os.environ.get("DLIO_BN_COOP", "1") != "0"]


def bn_coop_ok(N, HW):
    """the cooperative one-launch BatchNorm kernels (csrc/bn_small.hip, large planes) take this geometry"""
    return _BN_COOP[0] and _SYNC_BN[0] is None and bool(lib.dlio_bn_coop_ok(N, HW))


def bn_coop_set_cus(cus):
    """CUs a cooperative BatchNorm launch sizes its grid for (0 = default); see dlio_bn_coop_set_cus"""
    check(lib.dlio_bn_coop_set_cus(int(cus)), "bn_coop_set_cus")


def bn_coop_set_mode(oneshot):
    """3 (default): one item per workgroup where concurrent launches cannot fill an XCD with waiting workgroups, persistent
    otherwise; 2: persistent workgroups, one item at a time; 1: one item per workgroup (launches chained by _coop_enter);
    0: persistent, software-pipelined workgroups; -1: the default again (see dlio_bn_coop_set_mode)"""
    check(lib.dlio_bn_coop_set_mode(int(oneshot)), "bn_coop_set_mode")
    _COOP_TOKEN[0] = None              # (decided again from the mode now in force)


def bn_coop_gap_ok(N, HW):
    """the cooperative forward kernel can deliver plane averages (gap_out) for this geometry (fp32: any it takes -- the
    parts of a plane exchange their sums; the bf16 kernels ask dlio_bn_coop_gap_ok)"""
    return bn_coop_ok(N, HW)


# At most ONE cooperative BatchNorm launch in flight per device (DLIO_BN_COOP_TOKEN, on with the one-item-per-workgroup mode):
# every launch waits for the previous one -- whatever stream that was on -- and leaves its own event.  In that mode the
# partners of a channel are whichever workgroups the dispatcher starts next; two such launches (the two encoder streams) can
# fill an XCD's workgroup slots with waiting workgroups of BOTH while the partners they wait for are not yet dispatched --
# measured: one step in ~150 stalls until the spin limit, both launches at once.  One launch alone cannot: it waits with fewer
# than N * parts workgroups, an XCD holds more.  (The persistent mode draws every ticket from resident workgroups: no token.)
_COOP_TOKEN = [None]
_COOP_LAST = {}


def _coop_token_on():
    if _COOP_TOKEN[0] is None:
        env = os.environ.get("DLIO_BN_COOP_TOKEN")
        _COOP_TOKEN[0] = (env != "0") if env is not None else (lib.dlio_bn_coop_get_mode() == 1)
    return _COOP_TOKEN[0]


def _coop_enter(device):
    if not _coop_token_on():
        return
    dev = device.index if device.index is not None else torch._C._cuda_getDevice()
    last = _COOP_LAST.get(dev)
    if last is not None and last[0] != raw_stream():
        torch.cuda.current_stream().wait_event(last[1])


def _coop_exit(device):
    if not _coop_token_on():
        return
    dev = device.index if device.index is not None else torch._C._cuda_getDevice()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _COOP_LAST[dev] = (raw_stream(), ev)


def _coop_ws(N, C_, device):
    """(partial-sum slots, departure counters) of the cooperative BatchNorm kernels: one pair per (device, stream),
    initialised when allocated (slots = the library's "empty" pattern, counters = 0; the kernels restore both)"""
    key = (device.index if device.index is not None else torch._C._cuda_getDevice(), raw_stream())
    e = _COOP_WS.get(key)
    need_p, need_s = lib.dlio_bn_coop_ws_bytes(N, C_) // 8, C_ + 4
    if e is None or e[0].numel() < need_p or e[1].numel() < need_s:
        empty = lib.dlio_bn_coop_empty()
        e = (torch.full((max(need_p, 1 << 15),), empty - (1 << 64) if empty >= (1 << 63) else empty, dtype=torch.int64,
                        device=device),
             torch.zeros(max(need_s, 4096), dtype=torch.int32, device=device))
        _COOP_WS[key] = e
    return e


def bn_coop_fwd(x, x_ctot, x_coff, N, C_, C1, HW, set1, set2, eps, momentum, prm, y, y_ctot, y_coff, post_relu=True,
                residual=None, r_ctot=0, r_coff=0, r_aff=None, gap_out=None, gap_ctot=0, gap_coff=0, amax_out=None):
    """bn_small_fwd for large planes (N cooperating workgroups per channel)"""
    g2 = set2 if set2 is not None else (None, None, None, None)
    part, sync = _coop_ws(N, C_, x.device)
    _coop_enter(x.device)
    check(lib.dlio_bn_coop_fwd(_ptr(x), N, x_ctot, x_coff, C_, C1, HW, int(post_relu), _ptr(set1[0]), _ptr(set1[1]),
                               _ptr(set1[2]), _ptr(set1[3]), _ptr(g2[0]), _ptr(g2[1]), _ptr(g2[2]), _ptr(g2[3]), float(eps),
                               float(momentum), _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(residual),
                               r_ctot, r_coff, _ptr(r_aff[0]) if r_aff is not None else None,
                               _ptr(r_aff[1]) if r_aff is not None else None, _ptr(r_aff[2]) if r_aff is not None else None,
                               _ptr(y), y_ctot, y_coff, _ptr(gap_out), gap_ctot, gap_coff, _ptr(part), _ptr(sync),
                               _ptr(amax_out), _stream()),
          "bn_coop_fwd")
    _coop_exit(x.device)
    return prm


def bn_coop_bwd(dy, dy_ctot, dy_coff, x, x_ctot, x_coff, prm, beta1, beta2, dx1, dx2, dg1, db1, dg2, db2, accumulate, N,
                C_, C1, HW, post_relu=True, amax_out=None):
    """amax_out: a zeroed one-float tensor (amax_slot) that receives max |dx| (for the two-piece split kernels)"""
    part, sync = _coop_ws(N, C_, x.device)
    _coop_enter(x.device)
    check(lib.dlio_bn_coop_bwd(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]),
                               _ptr(beta1), _ptr(beta2), _ptr(dx1), _ptr(dx2), _ptr(dg1), _ptr(db1), _ptr(dg2), _ptr(db2),
                               int(accumulate), N, C_, C1, HW, int(post_relu), _ptr(part), _ptr(sync), _ptr(amax_out),
                               _stream()), "bn_coop_bwd")
    _coop_exit(x.device)


def bn_coop_pool_ok(N, H, W, SH):
    return bool(lib.dlio_bn_coop_pool_ok(N, H, W, SH))


def bn_coop_bwd_pool(dy, dy_ctot, dy_coff, pool, x, x_ctot, x_coff, prm, beta1, beta2, dx1, dx2, dg1, db1, dg2, db2,
                     accumulate, N, C_, C1, H, W, post_relu=True, amax_out=None):
    """bn_coop_bwd whose upstream gradient is x_scale * route(dy_pooled, idx) + x_add (+ dy, or None): pool = (dy_pooled, idx,
    x_scale, x_add, SH) from the SELayer + max-pool behind the block (the full-resolution gradient is never written)"""
    dyp, idx, xs, xa, SH = pool
    part, sync = _coop_ws(N, C_, x.device)
    _coop_enter(x.device)
    check(lib.dlio_bn_coop_bwd_pool(_ptr(dy), dy_ctot, dy_coff, _ptr(dyp), _ptr(idx), _ptr(xs), _ptr(xa), H, W, SH, _ptr(x),
                                    x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(beta1), _ptr(beta2),
                                    _ptr(dx1), _ptr(dx2), _ptr(dg1), _ptr(db1), _ptr(dg2), _ptr(db2), int(accumulate), N, C_,
                                    C1, int(post_relu), _ptr(part), _ptr(sync), _ptr(amax_out), _stream()), "bn_coop_bwd_pool")
    _coop_exit(x.device)


_AMAX = {}
_AMAX_N = 2048


def amax_slot(device):
    """a zeroed float on the device for a kernel's largest output magnitude.  Slots come from a ring of two halves, ONE RING
    PER (device, stream): the half about to be handed out is zeroed with one fill on the allocating stream when the
    allocation enters it, i.e. ordered in front of the producer that will write the slot (a ring shared by all streams had
    its fill race with the other encoder's producer).  The other half's slots are _AMAX_N / 2 allocations of this stream
    old by then; their last readers (the weight-gradient companion stream) were forked long before -- a training step takes
    ~15 slots per stream and joins all its streams at the optimizer."""
    key = (device.index if device.index is not None else torch._C._cuda_getDevice(), raw_stream())
    e = _AMAX.get(key)
    if e is None:
        e = _AMAX[key] = [torch.zeros(_AMAX_N, dtype=torch.float32, device=device), 0]
    buf, i = e
    half = _AMAX_N // 2
    if i % half == 0 and i >= half:                # (the first pass over the first half is the allocation's own zeros)
        buf[(i % _AMAX_N):(i % _AMAX_N) + half].zero_()
    e[1] = i + 1
    j = i % _AMAX_N
    return buf[j:j + 1]


_AMAX_KEEP = {}


def amax_slot_kept(device):
    """amax_slot for a value that is read again in the backward pass (the operand scale of a two-piece forward convolution,
    reused by its weight gradient): a ring of its own, four times as long -- a slot is zeroed again >= 4096 allocations of
    its stream later, hundreds of training steps"""
    key = (device.index if device.index is not None else torch._C._cuda_getDevice(), raw_stream())
    e = _AMAX_KEEP.get(key)
    n = 4 * _AMAX_N
    if e is None:
        e = _AMAX_KEEP[key] = [torch.zeros(n, dtype=torch.float32, device=device), 0]
    buf, i = e
    half = n // 2
    if i % half == 0 and i >= half:
        buf[(i % n):(i % n) + half].zero_()
    e[1] = i + 1
    j = i % n
    slot = buf[j:j + 1]
    slot._dlio_gen = (key, i)          # (ring, allocation number): amax_fresh() tells a consumer whether the slot still is its own
    return slot


def amax_fresh(slot):
    """a slot handed out by amax_slot_kept still holds its producer's value: the ring zeroes a slot again half a ring
    (2 x _AMAX_N = 4096 allocations of its stream) after it was handed out -- a tensor that outlived that many allocations (many
    train-mode forwards kept alive before their backward) must not be scaled by whatever the slot holds now.  Slots that did not
    come from the ring (a test's own one-float tensor) count as fresh."""
    gen = getattr(slot, "_dlio_gen", None)
    if gen is None:
        return True
    e = _AMAX_KEEP.get(gen[0])
    return e is not None and e[1] - gen[1] <= 2 * _AMAX_N


def conv3x3_h2_ok(desc):
    """the two-piece 3x3 kernel takes this launch (the producer / consumer kernel's sizes)"""
    return bool(lib.dlio_conv3x3_h2_ok(C.byref(desc)))


def conv3x3_h2_fwd(x, amax_x, wt, bias, y, desc, residual=None):
    """conv3x3_bx3_fwd on the two-piece fp16 split: amax_x = one-float tensor with max |x| (amax_slot filled by the producer
    of x), wt = conv_h2_prepped(w, mode)"""
    check(lib.dlio_conv3x3_h2_fwd(_ptr(x), _ptr(amax_x), _ptr(wt), _ptr(bias), _ptr(residual), _ptr(y), C.byref(desc),
                                  _stream()), "conv3x3_h2_fwd")
    return y


def bn_coop_errors():
    """number of cooperative BatchNorm workspaces whose launches hit the spin limit (0 = none); synchronises"""
    return sum(int(e[1][0].item() != 0) for e in _COOP_WS.values())


def bn_coop_check(fallback=True):
    """-> True when every cooperative BatchNorm launch so far found its partners.  Otherwise (a spin limit was hit: results
    of those launches are invalid) the workspaces are re-initialised and, with fallback, the cooperative kernels are
    switched off for the rest of the process -- the two-launch kernels of bn.hip take over.  Synchronises (one 4-byte read
    per workspace): call it where the caller synchronises anyway (TrainStep.check)."""
    bad = bn_coop_errors()
    if not bad:
        return True
    torch.cuda.synchronize()
    if os.environ.get("DLIO_BN_COOP_DEBUG"):
        for k, (part, sync) in _COOP_WS.items():
            e = lib.dlio_bn_coop_empty()
            e = e - (1 << 64) if e >= (1 << 63) else e
            print("[bn_coop] workspace", k, "header", sync[:4].tolist(), "nonzero counters", sync[4:].nonzero().flatten().tolist()[:8],
                  "non-empty slots", (part != e).nonzero().flatten().tolist()[:16], flush=True)
    empty = lib.dlio_bn_coop_empty()
    for part, sync in _COOP_WS.values():
        part.fill_(empty - (1 << 64) if empty >= (1 << 63) else empty)
        sync.zero_()
    if fallback:
        _BN_COOP[0] = False
        import warnings
        warnings.warn("deeplio_amd: a cooperative BatchNorm launch did not find its partner workgroups (%d workspace(s)); "
                      "falling back to the two-launch BatchNorm kernels" % bad, RuntimeWarning)
    return False


def fire_planes(N, S, H, W, device):
    """storage of the split squeeze activation of a Fire block (dlio_bn_split16 -> dlio_fire_expand_fwd)"""
    return torch.empty(lib.dlio_fire_planes_bytes(N, S, H, W), dtype=torch.uint8, device=device)


def bn_split16(x, x_ctot, x_coff, gamma, beta, eps, momentum, running_mean, running_var, y, y_ctot, y_coff, planes,
               N, C_, H, W, training, post_relu=True, fmt=0, bound_out=None):
    """the squeeze BatchNorm (+ ReLU) of a Fire block: activated fp32 tensor y (or None) + the split planes -> prm [3][C]
    (train: batch statistics, SyncBN-aware like bn_train_apply; eval: running statistics).  bound_out (fmt 1): a one-float
    tensor that receives the bound on |y| the two-piece scale was taken from"""
    if fmt and (not training or _SYNC_BN[0] is not None):
        raise ValueError("the two-piece planes exist in train mode with local batch statistics only")
    if not training:
        prm = bn_eval_params(running_mean, running_var, gamma, eps)
        check(lib.dlio_bn_split16(_ptr(x), N, x_ctot, x_coff, C_, H, W, int(post_relu), _ptr(gamma), _ptr(beta),
                                  float(eps), float(momentum), None, None, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]),
                                  _ptr(y), y_ctot, y_coff, _ptr(planes), None, 0, 3, 1.0, None, _stream()), "bn_split16")
        return prm
    prm = torch.empty(3, C_, dtype=torch.float32, device=x.device)
    ws = _stats_ws(N, C_, H * W, x.device)

    def call(mode, scale):
        check(lib.dlio_bn_split16(_ptr(x), N, x_ctot, x_coff, C_, H, W, int(post_relu), _ptr(gamma), _ptr(beta),
                                  float(eps), float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(prm[0]),
                                  _ptr(prm[1]), _ptr(prm[2]), _ptr(y), y_ctot, y_coff, _ptr(planes), _ptr(ws), ws.numel(),
                                  mode + (16 if fmt else 0), float(scale), _ptr(bound_out) if fmt else None, _stream()),
              "bn_split16")
    sync = _SYNC_BN[0]
    if sync is None:
        call(0, 1.0)
    else:
        call(1, 1.0)
        sync[0](_partials_view(ws, N, C_, H * W))
        call(2, sync[1])
    return prm


def fire_expand_fwd(planes, w3t, w1t, bias3, bias1, y, N, S, H, W, E, y_ctot, y_coff, fmt=0):
    """y[:, y_coff : y_coff + E] = expand1x1, y[:, y_coff + E : y_coff + 2 E] = expand3x3 of the split squeeze
    activation (w3t / w1t = conv_bx3_prepped(w, 0) of the two layers; fmt 1: two-piece planes (bn_split16 fmt=1) and
    conv_h2_prepped weights)"""
    check(lib.dlio_fire_expand_fwd(_ptr(planes), _ptr(w3t), _ptr(w1t), _ptr(bias3), _ptr(bias1), _ptr(y), N, S, H, W, E,
                                   y_ctot, y_coff, int(fmt), _stream()), "fire_expand_fwd")
    return y


def fire_expand_fwd_stats(planes, w3t, w1t, bias3, bias1, y, N, S, H, W, E, y_ctot, y_coff, set1, set3, eps, momentum,
                          mean, invstd, scale, shift, fmt=0):
    """fire_expand_fwd + the train-mode BatchNorm statistics of both expand layers from the launch's own tile sums
    (set = (gamma, beta, running_mean, running_var)); mean / invstd / scale / shift: rows of 2 E floats"""
    nbytes = lib.dlio_fire_expand_stats_ws_bytes(N, H, W, E)
    ws = workspace(nbytes, y.device, slot=5)
    check(lib.dlio_fire_expand_fwd_stats(_ptr(planes), _ptr(w3t), _ptr(w1t), _ptr(bias3), _ptr(bias1), _ptr(y), N, S, H, W, E,
                                         y_ctot, y_coff, _ptr(set1[0]), _ptr(set1[1]), _ptr(set1[2]), _ptr(set1[3]),
                                         _ptr(set3[0]), _ptr(set3[1]), _ptr(set3[2]), _ptr(set3[3]), float(eps),
                                         float(momentum), _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(ws),
                                         ws.numel(), int(fmt), _stream()), "fire_expand_fwd_stats")
    return y


def bn_bwd_fused(dy, dy_ctot, dy_coff, x, x_ctot, x_coff, prm, beta, dx, dx_ctot, dx_coff, N, C_, HW,
                 pre_relu, post_relu, use_batch_stats, dgamma=None, dbeta=None, accumulate=False, amax_out=None):
    """BN backward (reductions + dx, dgamma, dbeta) in two launches (SyncBN: partials all-reduced in
    between, dgamma / dbeta from the local copy).  amax_out: a zeroed one-float tensor that receives max |dx|"""
    ws = _stats_ws(N, C_, HW, x.device)

    def call(phase, scale, local):
        check(lib.dlio_bn_bwd(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[1]),
                              _ptr(prm[2]), _ptr(beta), _ptr(dx), dx_ctot, dx_coff, _ptr(dgamma), _ptr(dbeta),
                              int(accumulate), N, C_, HW, int(pre_relu), int(post_relu), int(use_batch_stats),
                              _ptr(ws), ws.numel(), phase, float(scale), _ptr(local),
                              _ptr(amax_out) if phase != 1 else None, _stream()), "bn_bwd")
    sync = _SYNC_BN[0]
    if sync is None or not use_batch_stats:
        call(0, 1.0, None)
    else:
        call(1, 1.0, None)
        part = _partials_view(ws, N, C_, HW)
        local = part.clone()
        sync[0](part)
        call(2, sync[1], local)
    return dx


def bn_bwd_pool(dy_pool, idx, x, prm, beta, dx, sh, dgamma=None, dbeta=None, accumulate=False):
    """BatchNorm + ReLU backward with the gradient gathered from the pooled gradient / arg-max map of the fast-path pool
    behind it (dlio_bn_bwd_pool); x, dx contiguous [N, C, H, W]"""
    N, C_, H, W = x.shape
    OH, OW = dy_pool.shape[2], dy_pool.shape[3]
    ws = _stats_ws(N, C_, H * W, x.device)
    check(lib.dlio_bn_bwd_pool(_ptr(dy_pool), _ptr(idx), _ptr(x), _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _ptr(beta),
                               _ptr(dx), _ptr(dgamma), _ptr(dbeta), int(accumulate), N, C_, H, W, OH, OW, int(sh),
                               _ptr(ws), ws.numel(), _stream()), "bn_bwd_pool")
    return dx


def bn_eval_params(running_mean, running_var, gamma, eps):
    C_ = running_mean.numel()
    prm = torch.empty(3, C_, dtype=torch.float32, device=running_mean.device)
    check(lib.dlio_bn_eval_params(_ptr(running_mean), _ptr(running_var), _ptr(gamma), float(eps),
                                  C_, _ptr(prm[0]), _ptr(prm[1]), _ptr(prm[2]), _stream()),
          "bn_eval_params")
    return prm


def bn_apply(x, x_ctot, x_coff, prm, beta, y, y_ctot, y_coff, N, C_, HW, pre_relu, post_relu,
             residual=None, r_ctot=0, r_coff=0):
    check(lib.dlio_bn_apply(_ptr(x), x_ctot, x_coff, _ptr(prm[0]), _ptr(prm[2]), _ptr(beta),
                            _ptr(residual), r_ctot, r_coff, _ptr(y), y_ctot, y_coff, N, C_, HW,
                            int(pre_relu), int(post_relu), _stream()), "bn_apply")
    return y


def bn_bwd(dy, dy_ctot, dy_coff, x, x_ctot, x_coff, prm, beta, dx, dx_ctot, dx_coff, N, C_, HW,
           pre_relu, post_relu, use_batch_stats, dgamma=None, dbeta=None, accumulate=False):
    sums = torch.empty(2, C_, dtype=torch.float64, device=x.device)
    ws = _stats_ws(N, C_, HW, x.device)
    check(lib.dlio_bn_bwd_reduce(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]),
                                 _ptr(prm[1]), _ptr(prm[2]), _ptr(beta), N, C_, HW, int(pre_relu),
                                 int(post_relu), _ptr(sums[0]), _ptr(sums[1]), _ptr(dgamma), _ptr(dbeta),
                                 int(accumulate), _ptr(ws), ws.numel(), _stream()), "bn_bwd_reduce")
    check(lib.dlio_bn_bwd_apply(_ptr(dy), dy_ctot, dy_coff, _ptr(x), x_ctot, x_coff, _ptr(prm[0]),
                                _ptr(prm[1]), _ptr(prm[2]), _ptr(beta), _ptr(sums[0]),
                                _ptr(sums[1]), _ptr(dx), dx_ctot, dx_coff, None, None,
                                N, C_, HW, int(pre_relu), int(post_relu), int(use_batch_stats),
                                _stream()), "bn_bwd_apply")
    return dx


def chan_sum(x, N, ctot, coff, C_, HW, out=None, accumulate=False):
    if out is None:
        out = torch.empty(C_, dtype=torch.float32, device=x.device)
    ws = _stats_ws(N, C_, HW, x.device)
    check(lib.dlio_chan_sum(_ptr(x), N, ctot, coff, C_, HW, _ptr(out), int(accumulate), _ptr(ws),
                            ws.numel(), _stream()), "chan_sum")
    return out


# ----------------------------------------------------------------------------- pooling
def pool_out(size, k, s, p, ceil_mode):
    if ceil_mode:
        o = -(-(size + 2 * p - k) // s) + 1
        if (o - 1) * s >= size + p:  # last window must start inside input or left padding
            o -= 1
        return o
    return (size + 2 * p - k) // s + 1


def maxpool2d_fwd(x, k, sh, sw, ph, pw, ceil_mode=False, x_scale=None, want_idx=True):
    N, C_, H, W = x.shape
    OH, OW = pool_out(H, k, sh, ph, ceil_mode), pool_out(W, k, sw, pw, ceil_mode)
    y = torch.empty(N, C_, OH, OW, dtype=torch.float32, device=x.device)
    idx = torch.empty(N, C_, OH, OW, dtype=torch.uint8, device=x.device) if want_idx else None
    check(lib.dlio_maxpool2d_fwd(_ptr(x), _ptr(x_scale), _ptr(y), _ptr(idx), N, C_, H, W, OH, OW, k,
                                 sh, sw, ph, pw, _stream()), "maxpool2d_fwd")
    return y, idx


def maxpool2d_fwd_aff(x, aff, k, sh, sw, ph, pw):
    """the fast-path pool over max(0, (x - aff[0]) * aff[1] + aff[2]) per channel (apply-on-load; aff [3, C] contiguous)"""
    N, C_, H, W = x.shape
    OH, OW = pool_out(H, k, sh, ph, False), pool_out(W, k, sw, pw, False)
    if tuple(aff.shape) != (3, C_) or not aff.is_contiguous():
        raise ValueError("aff must be a contiguous [3, C] tensor")
    y = torch.empty(N, C_, OH, OW, dtype=torch.float32, device=x.device)
    idx = torch.empty(N, C_, OH, OW, dtype=torch.uint8, device=x.device)
    check(lib.dlio_maxpool2d_fwd_aff(_ptr(x), _ptr(aff), _ptr(y), _ptr(idx), N, C_, H, W, OH, OW, k, sh, sw, ph, pw,
                                     _stream()), "maxpool2d_fwd_aff")
    return y, idx


def maxpool2d_bwd(dy, idx, in_shape, k, sh, sw, ph, pw, x_scale=None, x_add=None, out=None):
    N, C_, H, W = in_shape
    OH, OW = dy.shape[2], dy.shape[3]
    dx = out if out is not None else torch.empty(N, C_, H, W, dtype=torch.float32, device=dy.device)
    check(lib.dlio_maxpool2d_bwd(_ptr(dy), _ptr(idx), _ptr(x_scale), _ptr(x_add), _ptr(dx), N, C_, H, W,
                                 OH, OW, k, sh, sw, ph, pw, _stream()), "maxpool2d_bwd")
    return dx


def pool_fast_path(H, W, OH, OW, k, sh, sw, ph, pw):
    return (k == 3 and sw == 2 and ph == 1 and pw == 1 and sh in (1, 2) and W % 4 == 0 and OW * 2 == W
            and OH == (H + 2 - 3) // sh + 1)


def maxpool2d_bwd_dot(dy, idx, x, k, sh, sw, ph, pw):
    """ds[n][c] = sum_hw scatter(dy) * x (fast-path shapes)"""
    N, C_, H, W = x.shape
    OH, OW = dy.shape[2], dy.shape[3]
    ds = torch.empty(N, C_, dtype=torch.float32, device=dy.device)
    check(lib.dlio_maxpool2d_bwd_dot(_ptr(dy), _ptr(idx), _ptr(x), _ptr(ds), N, C_, H, W, OH, OW, k, sh,
                                     sw, ph, pw, _stream()), "maxpool2d_bwd_dot")
    return ds


def plane_dot(a, b, div=None):
    """out[n, c] = sum_hw a * b / div[n, c] over contiguous [N, C, H, W] tensors"""
    N, C_, H, W = a.shape
    out = torch.empty(N, C_, dtype=torch.float32, device=a.device)
    check(lib.dlio_plane_dot(_ptr(a), _ptr(b), _ptr(div), _ptr(out), N * C_, H * W, _stream()), "plane_dot")
    return out


_PAIR_WS = {}


def pair_fuse_fc_fwd(a, b, mode, w, bias, act):
    """feat = gap(a) (+|-) gap(b) [N, C]; y = act(feat w^T + bias) [N, F] in one launch (a, b contiguous [N, C, H, W])"""
    N, C_, H, W = a.shape
    F_ = w.shape[0]
    key = (a.device.index if a.device.index is not None else torch._C._cuda_getDevice(), raw_stream())
    need = lib.dlio_pair_fuse_fc_ws_bytes(N, C_, F_)
    e = _PAIR_WS.get(key)
    if e is None or e[0].numel() < need or e[1].numel() < N:
        e = (torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=a.device),
             torch.zeros(max(N, 256), dtype=torch.int32, device=a.device))       # counters: zero once, the kernel restores them
        _PAIR_WS[key] = e
    feat = torch.empty(N, C_, dtype=torch.float32, device=a.device)
    y = torch.empty(N, F_, dtype=torch.float32, device=a.device)
    check(lib.dlio_pair_fuse_fc_fwd(_ptr(a), _ptr(b), N, C_, H * W, int(mode), _ptr(w), _ptr(bias), F_, int(act), _ptr(feat),
                                    _ptr(y), _ptr(e[0]), e[0].numel(), _ptr(e[1]), _stream()), "pair_fuse_fc_fwd")
    return feat, y


def pair_fuse_bwd(df, shape, mode):
    """df [N, C] -> (da, db) [N, C, H, W]: the gradient of the two plane averages and the add / sub in one launch"""
    N, C_, H, W = shape
    da = torch.empty(shape, dtype=torch.float32, device=df.device)
    db = torch.empty(shape, dtype=torch.float32, device=df.device)
    check(lib.dlio_pair_fuse_bwd(_ptr(df), _ptr(da), _ptr(db), N, C_, H * W, int(mode), _stream()), "pair_fuse_bwd")
    return da, db


def se_fc_ok(N, C_, R):
    """the SELayer's two fully connected layers have the one-launch kernels for this geometry"""
    return bool(lib.dlio_se_fc_ok(N, C_, R))


def se_fc_fwd(g, w1, w2):
    """g [N, C] plane averages -> (h [N, R] = relu(g w1^T), s [N, C] = sigmoid(h w2^T)) in one launch"""
    N, C_ = g.shape
    R = w1.shape[0]
    h = torch.empty(N, R, dtype=torch.float32, device=g.device)
    s = torch.empty(N, C_, dtype=torch.float32, device=g.device)
    check(lib.dlio_se_fc_fwd(_ptr(g), _ptr(w1), _ptr(w2), _ptr(h), _ptr(s), N, C_, R, _stream()), "se_fc_fwd")
    return h, s


def se_fc_bwd(ds, s, h, g, w1, w2, dw1, dw2, accumulate, dg_scale=1.0):
    """backward of se_fc_fwd in two launches: -> dg [N, C] (times dg_scale); dw1 / dw2 written or accumulated"""
    N, C_ = g.shape
    R = w1.shape[0]
    dz2 = torch.empty(N, C_, dtype=torch.float32, device=g.device)
    dz1 = torch.empty(N, R, dtype=torch.float32, device=g.device)
    dg = torch.empty(N, C_, dtype=torch.float32, device=g.device)
    check(lib.dlio_se_fc_bwd(_ptr(ds), _ptr(s), _ptr(h), _ptr(g), _ptr(w1), _ptr(w2), _ptr(dz2), _ptr(dz1), _ptr(dg),
                             float(dg_scale), _ptr(dw1), _ptr(dw2), int(accumulate), N, C_, R, _stream()), "se_fc_bwd")
    return dg


def gap_fwd(x, N, ctot, coff, C_, HW):
    out = torch.empty(N, C_, dtype=torch.float32, device=x.device)
    check(lib.dlio_gap_fwd(_ptr(x), ctot, coff, _ptr(out), N, C_, HW, _stream()), "gap_fwd")
    return out


def gap_bwd(dout, dx, N, C_, HW, accumulate=False):
    check(lib.dlio_gap_bwd(_ptr(dout), _ptr(dx), N, C_, HW, int(accumulate), _stream()), "gap_bwd")
    return dx


def chan_scale_fwd(x, s, y=None):
    N, C_, H, W = x.shape
    if y is None:
        y = torch.empty_like(x)
    check(lib.dlio_chan_scale_fwd(_ptr(x), _ptr(s), _ptr(y), N, C_, H * W, _stream()),
          "chan_scale_fwd")
    return y


def chan_scale_bwd(dy, x, s):
    N, C_, H, W = x.shape
    dx = torch.empty_like(x)
    ds = torch.empty(N, C_, dtype=torch.float32, device=x.device)
    check(lib.dlio_chan_scale_bwd(_ptr(dy), _ptr(x), _ptr(s), _ptr(dx), _ptr(ds), N, C_, H * W,
                                  _stream()), "chan_scale_bwd")
    return dx, ds


# ----------------------------------------------------------------------------- dense
def linear_fwd(x, w, b, act=ACT_NONE, addend=None, out=None, M=None, ldx=None, ldadd=0, ldy=None):
    _chk(w)
    N_, K = w.shape
    if M is None:
        M = x.numel() // K
    if ldx is None:
        ldx = K
    if out is None:
        out = torch.empty(M, N_, dtype=torch.float32, device=x.device)
    if ldy is None:
        ldy = N_
    if addend is not None and ldadd == 0:
        ldadd = N_
    check(lib.dlio_linear_fwd(_ptr(x), ldx, _ptr(w), _ptr(b), _ptr(addend), ldadd, _ptr(out), ldy,
                              M, N_, K, act, _stream()), "linear_fwd")
    return out


def act_bwd(dy, y, act, out=None):
    if out is None:
        out = torch.empty_like(dy)
    check(lib.dlio_act_bwd(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), act, _stream()), "act_bwd")
    return out


def linear_bwd_data(dz, w, M, out=None, lddz=None, lddx=None, accumulate=False):
    N_, K = w.shape
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=dz.device)
    nbytes = lib.dlio_linear_bwd_data_ws_bytes(M, N_, K)
    ws = workspace(nbytes, dz.device, slot=3) if nbytes else None
    check(lib.dlio_linear_bwd_data(_ptr(dz), N_ if lddz is None else lddz, _ptr(w), _ptr(out),
                                   K if lddx is None else lddx, M, N_, K, int(accumulate),
                                   _ptr(ws), ws.numel() if ws is not None else 0, _stream()),
          "linear_bwd_data")
    return out


def linear_bwd_weight(dz, x, M, N_, K, dw=None, db=None, want_bias=True, lddz=None, ldx=None,
                      accumulate=False):
    if dw is None:
        dw = torch.empty(N_, K, dtype=torch.float32, device=dz.device)
    if db is None and want_bias:
        db = torch.empty(N_, dtype=torch.float32, device=dz.device)
    check(lib.dlio_linear_bwd_weight(_ptr(dz), N_ if lddz is None else lddz, _ptr(x),
                                     K if ldx is None else ldx, _ptr(dw), _ptr(db), M, N_, K,
                                     int(accumulate), _stream()), "linear_bwd_weight")
    return dw, db


def abs_max(x, keep=True):
    """-> a one-float tensor with max |x| (x contiguous): the operand scale of a two-piece convolution over a tensor that
    comes without one"""
    slot = amax_slot_kept(x.device) if keep else amax_slot(x.device)
    check(lib.dlio_abs_max(_ptr(x), x.numel(), _ptr(slot), _stream()), "abs_max")
    return slot


def ew_binary(a, b, op, out=None, amax_out=None):
    """amax_out: a zeroed one-float tensor that receives max |out| (tensors of a multiple of four elements, 16-byte aligned)"""
    if out is None:
        out = torch.empty_like(a)
    check(lib.dlio_ew_binary(_ptr(a), _ptr(b), _ptr(out), a.numel(), op, _ptr(amax_out), _stream()), "ew_binary")
    return out


def seg_sum_fwd(x, groups, rows, cols):
    y = torch.empty(groups, cols, dtype=torch.float32, device=x.device)
    check(lib.dlio_seg_sum_fwd(_ptr(x), _ptr(y), groups, rows, cols, _stream()), "seg_sum_fwd")
    return y


def seg_sum_bwd(dy, groups, rows, cols):
    dx = torch.empty(groups, rows, cols, dtype=torch.float32, device=dy.device)
    check(lib.dlio_seg_sum_bwd(_ptr(dy), _ptr(dx), groups, rows, cols, _stream()), "seg_sum_bwd")
    return dx


def ew_scale(a, alpha, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(lib.dlio_ew_scale(_ptr(a), float(alpha), _ptr(out), a.numel(), _stream()), "ew_scale")
    return out


def copy2d(src, lds, dst, ldd, rows, cols, accumulate=False, src_off=0, dst_off=0):
    sp = C.c_void_p(src.data_ptr() + 4 * src_off)
    dp = C.c_void_p(dst.data_ptr() + 4 * dst_off)
    check(lib.dlio_copy2d(sp, lds, dp, ldd, rows, cols, int(accumulate), _stream()), "copy2d")
    return dst


def dropout_fwd(x, p, seed, offset):
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(lib.dlio_dropout_fwd(_ptr(x), _ptr(y), _ptr(mask), x.numel(), float(p), int(seed), int(offset), _stream()),
          "dropout_fwd")
    return y, mask


def dropout_bwd(dy, mask, p):
    dx = torch.empty_like(dy)
    check(lib.dlio_dropout_bwd(_ptr(dy), _ptr(mask), _ptr(dx), dy.numel(), float(p), _stream()),
          "dropout_bwd")
    return dx


def soft_fusion_ok(R, Fa, Fb):
    return bool(lib.dlio_soft_fusion_ok(R, Fa, Fb))


def soft_fusion_fwd(a, lda, b, ldb, R, Fa, Fb, w1, b1, w2, b2):
    """DeepLIOFusionSoft.forward as one launch: rows of a / b with strides lda / ldb -> (out [R, Fa + Fb], gate [R, Fa + Fb] =
    [s1 | s2])"""
    out = torch.empty(R, Fa + Fb, dtype=torch.float32, device=a.device)
    gate = torch.empty(R, Fa + Fb, dtype=torch.float32, device=a.device)
    check(lib.dlio_soft_fusion_fwd(_ptr(a), lda, _ptr(b), ldb, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out), _ptr(gate), R,
                                   Fa, Fb, _stream()), "soft_fusion_fwd")
    return out, gate


def soft_fusion_bwd(dout, a, lda, b, ldb, R, Fa, Fb, gate, w1, w2, dw1, db1, dw2, db2, accumulate):
    da = torch.empty(R, Fa, dtype=torch.float32, device=dout.device)
    db = torch.empty(R, Fb, dtype=torch.float32, device=dout.device)
    check(lib.dlio_soft_fusion_bwd(_ptr(dout), _ptr(a), lda, _ptr(b), ldb, _ptr(gate), _ptr(w1), _ptr(w2), _ptr(da), _ptr(db),
                                   _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), R, Fa, Fb, int(accumulate), _stream()),
          "soft_fusion_bwd")
    return da, db


def heads_ok(R, K, ldx):
    return bool(lib.dlio_heads_ok(R, K, ldx))


def heads_fwd(x, ldx, R, K, wp, bp, wo, bo, p, seed, offset):
    """dropout(p) + fc_pos + fc_ori over rows of x (row stride ldx) -> (pos [R, 3], ori [R, 3], mask [R, K] u8 | None)"""
    pos = torch.empty(R, 3, dtype=torch.float32, device=x.device)
    ori = torch.empty(R, 3, dtype=torch.float32, device=x.device)
    mask = torch.empty(R, K, dtype=torch.uint8, device=x.device) if p > 0. else None
    check(lib.dlio_heads_fwd(_ptr(x), ldx, _ptr(mask), _ptr(wp), _ptr(bp), _ptr(wo), _ptr(bo), _ptr(pos), _ptr(ori), R, K,
                             float(p), int(seed), int(offset), _stream()), "heads_fwd")
    return pos, ori, mask


def heads_bwd(dpos, dori, x, ldx, mask, wp, wo, dx, lddx, dwp, dbp, dwo, dbo, R, K, p, accumulate):
    check(lib.dlio_heads_bwd(_ptr(dpos), _ptr(dori), _ptr(x), ldx, _ptr(mask), _ptr(wp), _ptr(wo), _ptr(dx), lddx, _ptr(dwp),
                             _ptr(dbp), _ptr(dwo), _ptr(dbo), R, K, float(p), int(accumulate), _stream()), "heads_bwd")


def nonfinite_flag(x, flag):
    check(lib.dlio_nonfinite_flag(_ptr(x), x.numel(), _ptr(flag), _stream()), "nonfinite_flag")


# ----------------------------------------------------------------------------- rnn
def _rnn_ws(T, B, H, device):
    return workspace(lib.dlio_rnn_ws_bytes(T, B, H), device, slot=2)


def _off(t, off_floats):
    return C.c_void_p(t.data_ptr() + 4 * off_floats)


def lstm_seq_fwd(gx, w_hh, b_hh, h0, c0, hs, hs_off, ldhs, cs, hp, gates, hT, cT, T, B, H, rst,
                 rsb, reverse):
    ws = _rnn_ws(T, B, H, gx.device)
    check(lib.dlio_lstm_seq_fwd(_ptr(gx), _ptr(w_hh), _ptr(b_hh), _ptr(h0), _ptr(c0),
                                _off(hs, hs_off), ldhs, _ptr(cs), _ptr(hp), _ptr(gates), _ptr(hT),
                                _ptr(cT), T, B, H, rst, rsb, int(reverse), _ptr(ws), ws.numel(),
                                _stream()), "lstm_seq_fwd")


def lstm_seq_bwd(dhs, dhs_off, lddhs, dhT, dcT, gates, cs, c0, w_hh, dgates, dh0, dc0, T, B, H,
                 rst, rsb, reverse):
    ws = _rnn_ws(T, B, H, gates.device)
    dp = None if dhs is None else _off(dhs, dhs_off)
    check(lib.dlio_lstm_seq_bwd(dp, lddhs, _ptr(dhT), _ptr(dcT), _ptr(gates), _ptr(cs), _ptr(c0),
                                _ptr(w_hh), _ptr(dgates), _ptr(dh0), _ptr(dc0), T, B, H, rst, rsb,
                                int(reverse), _ptr(ws), ws.numel(), _stream()), "lstm_seq_bwd")


def lstm_layer_ok(T, B, I, H, D):
    """the both-directions-per-launch streamed LSTM layer (csrc/lstm_stream.hip) takes this geometry"""
    return bool(lib.dlio_lstm_layer_ok(T, B, I, H, D))


def lstm_layer_fwd(x, ldx, w, hs, ldhs, T, B, I, H, D):
    """one LSTM layer, D directions, T steps from the zero state.  x [B*T rows (b T + t)][ldx]; w = per direction (w_ih, w_hh,
    b_ih, b_hh); hs [rows][ldhs] receives direction d in columns d H ..  -> (cs, hp, gates) saved for lstm_layer_bwd"""
    rows = B * T
    cs = torch.empty(D, rows, H, dtype=torch.float32, device=x.device)
    hp = torch.empty(D, rows, H, dtype=torch.float32, device=x.device)
    gates = torch.empty(D, rows, 4 * H, dtype=torch.float32, device=x.device)
    ws = workspace(lib.dlio_lstm_layer_ws_bytes(T, B, I, H, D), x.device, slot=2)
    w1 = w[1] if D == 2 else (None, None, None, None)
    check(lib.dlio_lstm_layer_fwd(_ptr(x), ldx, _ptr(w[0][0]), _ptr(w[0][1]), _ptr(w[0][2]), _ptr(w[0][3]), _ptr(w1[0]),
                                  _ptr(w1[1]), _ptr(w1[2]), _ptr(w1[3]), _ptr(hs), ldhs, _ptr(cs), _ptr(hp), _ptr(gates), T, B,
                                  I, H, D, _ptr(ws), ws.numel(), _stream()), "lstm_layer_fwd")
    return cs, hp, gates


def lstm_layer_wgrad(dgates, x, ldx, hp, dws, accumulate, T, B, I, H, D):
    """the layer's weight + bias gradients from lstm_layer_bwd(.., dws=None)'s dgates (one launch)"""
    g1 = dws[1] if D == 2 else (None, None, None, None)
    check(lib.dlio_lstm_layer_wgrad(_ptr(dgates), _ptr(x), ldx, _ptr(hp), _ptr(dws[0][0]), _ptr(dws[0][1]), _ptr(dws[0][2]),
                                    _ptr(dws[0][3]), _ptr(g1[0]), _ptr(g1[1]), _ptr(g1[2]), _ptr(g1[3]), int(accumulate), T, B, I,
                                    H, D, _stream()), "lstm_layer_wgrad")


def lstm_layer_bwd(dhs, lddhs, x, ldx, hp, gates, cs, w, dws, accumulate, dx, lddx, T, B, I, H, D):
    """w = per direction (w_ih, w_hh); dws = per direction (dw_ih, dw_hh, db_ih, db_hh) written (or accumulated into), or None:
    the data path only (lstm_layer_wgrad takes the returned dgates); dx [rows][lddx] or None"""
    rows = B * T
    dgates = torch.empty(D, rows, 4 * H, dtype=torch.float32, device=x.device)
    ws = workspace(lib.dlio_lstm_layer_ws_bytes(T, B, I, H, D), x.device, slot=2)
    w1 = w[1] if D == 2 else (None, None)
    if dws is None:
        dws = [(None, None, None, None)] * D
    g1 = dws[1] if D == 2 else (None, None, None, None)
    check(lib.dlio_lstm_layer_bwd(_ptr(dhs), lddhs, _ptr(x), ldx, _ptr(hp), _ptr(gates), _ptr(cs), _ptr(w[0][0]), _ptr(w[0][1]),
                                  _ptr(w1[0]), _ptr(w1[1]), _ptr(dgates), _ptr(dws[0][0]), _ptr(dws[0][1]), _ptr(dws[0][2]),
                                  _ptr(dws[0][3]), _ptr(g1[0]), _ptr(g1[1]), _ptr(g1[2]), _ptr(g1[3]), int(accumulate),
                                  _ptr(dx), lddx, T, B, I, H, D, _ptr(ws), ws.numel(), _stream()), "lstm_layer_bwd")
    return dgates


def gru_seq_fwd(gx, w_hh, b_hh, h0, hs, hs_off, ldhs, hp, gates, hT, T, B, H, rst, rsb, reverse):
    ws = _rnn_ws(T, B, H, gx.device)
    check(lib.dlio_gru_seq_fwd(_ptr(gx), _ptr(w_hh), _ptr(b_hh), _ptr(h0), _off(hs, hs_off), ldhs,
                               _ptr(hp), _ptr(gates), _ptr(hT), T, B, H, rst, rsb, int(reverse),
                               _ptr(ws), ws.numel(), _stream()), "gru_seq_fwd")


def gru_seq_bwd(dhs, dhs_off, lddhs, dhT, gates, hp, w_hh, dgx, dgh, dh0, T, B, H, rst, rsb,
                reverse):
    ws = _rnn_ws(T, B, H, gates.device)
    dp = None if dhs is None else _off(dhs, dhs_off)
    check(lib.dlio_gru_seq_bwd(dp, lddhs, _ptr(dhT), _ptr(gates), _ptr(hp), _ptr(w_hh), _ptr(dgx),
                               _ptr(dgh), _ptr(dh0), T, B, H, rst, rsb, int(reverse), _ptr(ws),
                               ws.numel(), _stream()), "gru_seq_bwd")


# ----------------------------------------------------------------------------- pose
def se3_chain_fwd(t, w, order=0, status=None):
    _chk(t)
    _chk(w)
    B, S, _ = t.shape
    p = torch.empty(B, S, 3, dtype=torch.float32, device=t.device)
    q = torch.empty(B, S, 4, dtype=torch.float32, device=t.device)
    R = torch.empty(B, S, 9, dtype=torch.float32, device=t.device)
    check(lib.dlio_se3_chain_fwd(_ptr(t), _ptr(w), _ptr(p), _ptr(q), _ptr(R), _ptr(status), B, S,
                                 order, _stream()), "se3_chain_fwd")
    return p, q, R


def se3_chain_bwd(t, w, R, dp, dq, order=0):
    B, S, _ = t.shape
    dt = torch.empty_like(t)
    dw = torch.empty_like(w)
    check(lib.dlio_se3_chain_bwd(_ptr(t), _ptr(w), _ptr(R), _ptr(dp), _ptr(dq), _ptr(dt), _ptr(dw),
                                 B, S, order, _stream()), "se3_chain_bwd")
    return dt, dw


def so3_project(R):
    """R [..., 3, 3] (or [..., 9]) -> (Q, valid): SO3.normalize of liegroups without an SVD"""
    _chk(R)
    n = R.numel() // 9
    Q = torch.empty_like(R)
    valid = torch.empty(n, dtype=torch.int32, device=R.device)
    check(lib.dlio_so3_project(_ptr(R), _ptr(Q), _ptr(valid), n, _stream()), "so3_project")
    return Q, valid


def so3_project_bwd(R, G):
    _chk(R)
    _chk(G)
    dR = torch.empty_like(R)
    check(lib.dlio_so3_project_bwd(_ptr(R), _ptr(G), _ptr(dR), R.numel() // 9, _stream()), "so3_project_bwd")
    return dR


def _ptr_array(ts):
    arr = (C.c_void_p * 4)()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def pose_loss_fwd(preds, gts, sx, sq, beta, mode):
    n = (C.c_int32 * 4)(*[0 if p is None else p.numel() for p in preds])
    dev = next(p for p in preds if p is not None).device
    out = torch.empty(5, dtype=torch.float32, device=dev)
    check(lib.dlio_pose_loss_fwd(_ptr_array(preds), _ptr_array(gts), n, _ptr(sx), _ptr(sq),
                                 float(beta), mode, _ptr(out), _stream()), "pose_loss_fwd")
    return out


def pose_loss_bwd(preds, gts, sx, sq, beta, mode, out, gscale):
    n = (C.c_int32 * 4)(*[0 if p is None else p.numel() for p in preds])
    dpreds = [None if p is None else torch.empty_like(p) for p in preds]
    dev = out.device
    dsx = torch.empty((), dtype=torch.float32, device=dev) if (mode & 1) == 0 else None
    dsq = torch.empty((), dtype=torch.float32, device=dev) if (mode & 1) == 0 else None
    check(lib.dlio_pose_loss_bwd(_ptr_array(preds), _ptr_array(gts), n, _ptr(sx), _ptr(sq),
                                 float(beta), mode, _ptr(out), _ptr(gscale), _ptr_array(dpreds),
                                 _ptr(dsx), _ptr(dsq), _stream()), "pose_loss_bwd")
    return dpreds, dsx, dsq


def pose_tail_fwd(t, w, gt_f2f, gt_f2g, g0, g1, terms, sx, sq, beta, mode, order, status=None, nonfinite=None):
    """SE(3) chain + criterion on slices read in place (dlio_pose_tail_fwd) -> out [5] (loss, Lt, Lw, Lp, Lq), p, q, R_all"""
    for x in (t, w, gt_f2f, gt_f2g):
        _chk(x)
    B, S, _ = t.shape
    assert tuple(w.shape) == (B, S, 3) and tuple(gt_f2f.shape) == (B, S, 6) and tuple(gt_f2g.shape) == (B, S, 7)
    dev = t.device
    p = torch.empty(B, S, 3, dtype=torch.float32, device=dev)
    q = torch.empty(B, S, 4, dtype=torch.float32, device=dev)
    R = torch.empty(B, S, 9, dtype=torch.float32, device=dev)
    out = torch.empty(5, dtype=torch.float32, device=dev)
    check(lib.dlio_pose_tail_fwd(_ptr(t), _ptr(w), _ptr(gt_f2f), _ptr(gt_f2g), B, S, g0, g1, terms, _ptr(sx), _ptr(sq),
                                 float(beta), mode, order, _ptr(p), _ptr(q), _ptr(R), _ptr(status), _ptr(nonfinite), _ptr(out),
                                 _stream()), "pose_tail_fwd")
    return out, p, q, R


def pose_tail_bwd(t, w, gt_f2f, gt_f2g, g0, g1, terms, sx, sq, beta, mode, order, p, q, R, out, gscale, dsx=None, dsq=None,
                  acc_hyper=False):
    """-> dt, dw [B,S,3] (dlio_pose_tail_bwd); dsx / dsq are written (added to with acc_hyper) when given"""
    B, S, _ = t.shape
    ws = torch.empty(lib.dlio_pose_tail_ws_floats(B, S, g0, g1), dtype=torch.float32, device=t.device)
    dt, dw = torch.empty_like(t), torch.empty_like(w)
    check(lib.dlio_pose_tail_bwd(_ptr(t), _ptr(w), _ptr(gt_f2f), _ptr(gt_f2g), B, S, g0, g1, terms, _ptr(sx), _ptr(sq),
                                 float(beta), mode, order, _ptr(p), _ptr(q), _ptr(R), _ptr(out), _ptr(gscale), _ptr(ws),
                                 _ptr(dt), _ptr(dw), _ptr(dsx), _ptr(dsq), int(acc_hyper), _stream()), "pose_tail_bwd")
    return dt, dw


# ----------------------------------------------------------------------------- batch prep
def pair_stack(images, comb, c_split):
    """images [B,F,Ctot,H,W], comb int32 [S,2] (device) -> xyz [B,S,2,c_split,H,W], normals [...]"""
    _chk(images)
    B, F_, Ctot, H, W = images.shape
    S = comb.shape[0]
    xyz = torch.empty(B, S, 2, c_split, H, W, dtype=torch.float32, device=images.device)
    nrm = torch.empty(B, S, 2, Ctot - c_split, H, W, dtype=torch.float32, device=images.device)
    check(lib.dlio_pair_stack(_ptr(images), _ptr(comb), _ptr(xyz), _ptr(nrm), B, F_, Ctot, c_split, H, W,
                              S, _stream()), "pair_stack")
    return xyz, nrm


def gt_relative(gts, comb, flag=None):
    """gts [B,F,15], comb int32 [S,2] -> f2f [B,S,6], f2g [B,S,7]"""
    _chk(gts)
    B, F_, _ = gts.shape
    S = comb.shape[0]
    f2f = torch.empty(B, S, 6, dtype=torch.float32, device=gts.device)
    f2g = torch.empty(B, S, 7, dtype=torch.float32, device=gts.device)
    check(lib.dlio_gt_relative(_ptr(gts), _ptr(comb), _ptr(f2f), _ptr(f2g), _ptr(flag), B, F_, S,
                               _stream()), "gt_relative")
    return f2f, f2g


# ----------------------------------------------------------------------------- lidar scan
def scan_project(points, remissions, H, W, fov_up, fov_down):
    """points [N,3] f32 (device), remissions [N] or None -> dict of the LaserScan projection
    attributes (laserscan.py:122-185), all on the device."""
    _chk(points)
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("points must be [N,3]")
    if remissions is not None:
        _chk(remissions)
        if remissions.numel() != points.shape[0]:
            raise ValueError("remissions must have one value per point")
    N, dev = points.shape[0], points.device
    i32, f32 = torch.int32, torch.float32
    out = dict(proj_x=torch.empty(N, dtype=i32, device=dev), proj_y=torch.empty(N, dtype=i32, device=dev),
               unproj_range=torch.empty(N, dtype=f32, device=dev),
               proj_range=torch.empty(H, W, dtype=f32, device=dev),
               proj_xyz=torch.empty(H, W, 3, dtype=f32, device=dev),
               proj_remission=torch.empty(H, W, dtype=f32, device=dev),
               proj_idx=torch.empty(H, W, dtype=i32, device=dev),
               proj_mask=torch.empty(H, W, dtype=i32, device=dev))
    nb = lib.dlio_scan_project_ws_bytes(H, W)
    ws = workspace(nb, dev, slot=3)
    check(lib.dlio_scan_project(_ptr(points), _ptr(remissions), N, H, W, float(fov_up), float(fov_down),
                                _ptr(out["proj_x"]), _ptr(out["proj_y"]), _ptr(out["unproj_range"]),
                                _ptr(out["proj_range"]), _ptr(out["proj_xyz"]), _ptr(out["proj_remission"]),
                                _ptr(out["proj_idx"]), _ptr(out["proj_mask"]), _ptr(ws), ws.numel(), _stream()),
          "scan_project")
    return out


def scan_normals(proj_xyz, proj_range):
    _chk(proj_xyz)
    _chk(proj_range)
    H, W = proj_range.shape
    out = torch.empty(H, W, 3, dtype=torch.float32, device=proj_xyz.device)
    check(lib.dlio_scan_normals(_ptr(proj_xyz), _ptr(proj_range), _ptr(out), H, W, _stream()), "scan_normals")
    return out


def velo_image(proj_xyz, proj_remission, normals, proj_range, max_depth, channels, mean=None,
               crop_top=0, crop_left=0):
    """-> [len(channels), H-2*crop_top, W-2*crop_left] (kitti.py:83-97 + :345-364)"""
    for t in (proj_xyz, proj_remission, normals, proj_range):
        _chk(t)
    H, W = proj_range.shape
    ch = (C.c_int32 * len(channels))(*[int(c) for c in channels])
    mn = None
    if mean is not None:
        if len(mean) != 8:
            raise ValueError("mean must have 8 entries (one per image channel)")
        mn = (C.c_float * 8)(*[float(m) for m in mean])
    out = torch.empty(len(channels), H - 2 * crop_top, W - 2 * crop_left, dtype=torch.float32,
                      device=proj_xyz.device)
    check(lib.dlio_velo_image(_ptr(proj_xyz), _ptr(proj_remission), _ptr(normals), _ptr(proj_range),
                              float(max_depth), ch, mn, len(channels), H, W, crop_top, crop_left, _ptr(out),
                              _stream()), "velo_image")
    return out


# ----------------------------------------------------------------------------- optimizer
def optim_set_max_blocks(blocks):
    """workgroups the optimizer sweeps may use (0: default)"""
    check(lib.dlio_optim_set_max_blocks(int(blocks)), "optim_set_max_blocks")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    weights_changed()
    check(lib.dlio_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), float(beta1),
                             float(beta2), float(eps), float(wd), int(step), float(grad_scale),
                             _stream()), "adam_step")


def sgd_step(p, g, buf, lr, momentum, wd, step, grad_scale=1.0):
    weights_changed()
    check(lib.dlio_sgd_step(_ptr(p), _ptr(g), _ptr(buf), p.numel(), float(lr), float(momentum),
                            float(wd), int(step), float(grad_scale), _stream()), "sgd_step")


def rmsprop_step(p, g, square_avg, buf, grad_avg, lr, alpha, eps, wd, momentum, grad_scale=1.0):
    weights_changed()
    check(lib.dlio_rmsprop_step(_ptr(p), _ptr(g), _ptr(square_avg), _ptr(buf), _ptr(grad_avg), p.numel(),
                                float(lr), float(alpha), float(eps), float(wd), float(momentum),
                                float(grad_scale), _stream()), "rmsprop_step")


def adadelta_step(p, g, square_avg, acc_delta, lr, rho, eps, wd, grad_scale=1.0):
    weights_changed()
    check(lib.dlio_adadelta_step(_ptr(p), _ptr(g), _ptr(square_avg), _ptr(acc_delta), p.numel(), float(lr),
                                 float(rho), float(eps), float(wd), float(grad_scale), _stream()),
          "adadelta_step")


def sumsq(g):
    out = torch.empty(1, dtype=torch.float64, device=g.device)
    check(lib.dlio_sumsq(_ptr(g), g.numel(), _ptr(out), _stream()), "sumsq")
    return out


# ----------------------------------------------------------------------------- profiling
def prof_enable(on):
    """on: False / True (all kinds) or a bit mask of kernel kinds (include/deeplio_hip.h: bit 0 fp32
    multi-tap forward + data gradient, 1 other weight gradients, 2 1x1 forward + data gradient,
    3 split-bf16 3x3 forward + data gradient, 4 3x3 weight gradient, 5 1x1 weight gradient,
    6-9 BatchNorm statistics / apply / backward reductions / backward apply, 10 pools)"""
    lib.dlio_prof_enable(0xffff if on is True else int(on))


def prof_sample(stride):
    """time every stride-th launch of each enabled kind only (1 = every launch)"""
    check(lib.dlio_prof_sample(int(stride)), "prof_sample")


def prof_reset():
    lib.dlio_prof_reset()


def prof_release():
    lib.dlio_prof_release()


def prof_collect(kind):
    ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    check(lib.dlio_prof_collect(kind, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)),
          "prof_collect")
    return dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)
