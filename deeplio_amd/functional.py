"""torch.autograd.Function wrappers that compose the C-ABI kernels into the layers of the
DeepLIO hot path.  Every forward/backward below is a sequence of libdeeplio_hip launches on
the current HIP stream; torch supplies storage, views and the autograd tape only.

Activations are fp32 NCHW.  Concatenations (Fire's expand1x1 || expand3x3) are never
materialised by a copy: both convolutions write channel slices of one buffer.
"""
import contextlib
import os

import torch
from torch.autograd import Function

from . import ops

_E = torch.empty


def _new(shape, like):
    return _E(shape, dtype=torch.float32, device=like.device)


_SINK = [True]
_PLANE_BN = [True]            # (a module switch for tests / ablations; the environment knob is gone)
# stem: pool1's backward folded into the stem's BatchNorm backward (ConvBnActPoolFn)
_POOL_BN_BWD = [True]


def set_grad_sink(on):
    """When on (default), parameter gradients are accumulated by the kernels straight into an
    existing `param.grad` buffer (the optimizer's flat gradient view) and autograd receives None
    for them: no per-parameter `+=` launches, no extra pass over the gradients."""
    _SINK[0] = bool(on)


# ------------------------------------------------------------------------------ auxiliary streams
# Independent parts of the step run on their own HIP streams: the IMU branch, the second
# siamese encoder (nets.py) and -- in backward -- every weight-gradient kernel, which nothing
# downstream in the tape consumes.  The streams live in one registry so that they can be joined
# into the caller's stream when a backward pass ends (autograd only knows about streams that
# tensors it routes were produced on; gradients accumulated in place by the sink are invisible
# to it) and again, explicitly, by FlatOptimizer.step()/GradSync.
_AUX = {}                   # (device index, name) -> torch.cuda.Stream
_JOIN_PENDING = [False]
# weight-gradient fork: slower when first tried (43.5 vs 41.9 ms/step, staged wgrad kernels that filled
# the chip); with the direct kernels it is 33.2 vs 33.9 ms/step -> on
_WGRAD_FORK = [os.environ.get("DLIO_WGRAD_STREAM", "1") != "0"]


class on_stream:
    """`with torch.cuda.stream(s)` without its Python layers (current_stream() + two Stream objects +
    device guards: ~25 us per use, ~120 uses per step): set torch's current stream through the C entry
    points and restore the previous one"""
    __slots__ = ("s", "prev")

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = torch._C._cuda_getCurrentStream(self.s.device_index)
        torch._C._cuda_setStream(stream_id=self.s.stream_id, device_index=self.s.device_index,
                                 device_type=self.s.device_type)
        return self.s

    def __exit__(self, *exc):
        p = self.prev
        torch._C._cuda_setStream(stream_id=p[0], device_index=p[1], device_type=p[2])
        return False


def current_stream_obj(device_index=None):
    """torch.cuda.current_stream() through the C getter"""
    if device_index is None:
        device_index = torch._C._cuda_getDevice()
    p = torch._C._cuda_getCurrentStream(device_index)
    return torch.cuda.Stream(stream_id=p[0], device_index=p[1], device_type=p[2])


def aux_stream(device, name):
    key = (device.index if device.index is not None else torch.cuda.current_device(), name)
    s = _AUX.get(key)
    if s is None:
        s = _AUX[key] = torch.cuda.Stream(device=device)
    return s


_ASSIGNED = set()
ASSIGN_REPORT = {}          # device index -> what assign_streams found (bench.py prints it)


def assign_streams(device, force=False):
    """Create the step's auxiliary streams so that its four HEAVY streams -- the current one (first encoder), 'encoder2' and
    their two weight-gradient companions -- are served by four different hardware queues.  The HIP runtime hands a new stream
    the least used of its (4) hardware queues, i.e. the outcome depends on every stream the process created before (RCCL's,
    another library's, a test's): two heavy streams on one queue run one after the other and the step takes 20.9-21.3 instead
    of 19.0 ms.  Candidates come from torch's stream pool and are sorted by queue with dlio_streams_share_queue;
    the light streams (IMU branch, reverse directions of the two bidirectional RNNs, the data-parallel exchange) each get a
    stream on the queue of a heavy one that is idle -- or has slack -- while they run.  Once per device (TrainStep.__init__ calls it); DLIO_ASSIGN_STREAMS=0 leaves the
    streams to their order of first use."""
    if not (torch.cuda.is_available() and device.type == "cuda"):
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if (idx in _ASSIGNED and not force) or os.environ.get("DLIO_ASSIGN_STREAMS", "1") == "0":
        return None
    with torch.cuda.device(idx):            # (the probe creates events and launches on the CURRENT device)
        table = _assign_streams_on(idx, force)
    _ASSIGNED.add(idx)                      # (only a probe that ran to its end settles the device: an exception above leaves
    return table                            #  the next caller free to try again)


def _assign_streams_on(idx, force):
    main = torch.cuda.current_stream(idx)
    if not force and any(k[0] == idx for k in _AUX):
        return None                  # somebody has run on this device already: leave its streams alone
    for k in [k for k in _AUX if k[0] == idx]:
        del _AUX[k]
    cands, used, memo = [], set(), {}

    def share(a, b):
        k = (a.cuda_stream, b.cuda_stream)
        if k not in memo:
            memo[k] = memo[(k[1], k[0])] = ops.streams_share_queue(a, b)
        return memo[k]

    def cand(j):
        while len(cands) <= j:
            cands.append(torch.cuda.Stream(device=idx))
        return cands[j]

    def pick(avoid, prefer=None, limit=12):
        """an unused candidate that shares no queue with `avoid` (and the queue of `prefer`, if given)"""
        for j in range(limit):
            if j in used:
                continue
            c = cand(j)
            if any(share(h, c) for h in avoid):
                continue
            if prefer is not None and not share(prefer, c):
                continue
            used.add(j)
            return c
        return None

    def any_unused():
        j = next(j for j in range(64) if j not in used)
        used.add(j)
        return cand(j)

    heavy = [main]
    for _ in range(3):
        heavy.append(pick(heavy) or any_unused())        # (fewer than four queues: GPU_MAX_HW_QUEUES < 4)
    enc2, wg0, wg2 = heavy[1:]
    distinct = sum(1 for i, h in enumerate(heavy) if not any(share(h, g) for g in heavy[:i]))
    ASSIGN_REPORT[idx] = {"heavy_streams_on_distinct_queues": distinct, "candidates_probed": len(cands)}
    if distinct < 4:
        import warnings
        warnings.warn("deeplio_amd: the step's four heavy HIP streams share hardware queues (%d distinct found; "
                      "GPU_MAX_HW_QUEUES=%s): two of them on one queue cost ~10 %% of the step"
                      % (distinct, os.environ.get("GPU_MAX_HW_QUEUES", "default 4")), RuntimeWarning)
    imu = pick([main, enc2], prefer=wg2) or pick([main, enc2]) or any_unused()
    rd_imu = pick([main, enc2, imu], prefer=wg0) or pick([main, imu]) or any_unused()
    rd_main = pick([main, wg0, wg2], prefer=enc2) or pick([main]) or any_unused()
    # data parallel: the stream the overlapped gradient all-reduce is issued on (dist.GradSync.reduce_tail_async) -- beside a
    # weight-gradient companion, never on an encoder's queue (a collective there would hold that encoder for its ~3 ms)
    comm = pick([main, enc2], prefer=wg0) or pick([main, enc2]) or any_unused()
    # host-fed training: the stream the NEXT batch's H2D copies + DataCombiCreater kernels are issued on (a barrier packet
    # behind a ~2.5 ms PCIe copy holds its hardware queue: beside a weight-gradient companion, which is idle during the
    # forward pass the copy runs under -- never on an encoder's queue)
    feed = pick([main, enc2], prefer=wg0) or pick([main, enc2]) or any_unused()
    table = {"encoder2": enc2, "wgrad@%x" % main.cuda_stream: wg0, "wgrad@%x" % enc2.cuda_stream: wg2, "imu": imu,
             "rnndir@%x" % imu.cuda_stream: rd_imu, "rnndir@%x" % main.cuda_stream: rd_main, "comm": comm, "feed": feed}
    for name, st in table.items():
        _AUX[(idx, name)] = st
    return table


def join_aux_streams():
    """current stream waits for everything issued so far on every auxiliary stream of its device"""
    _JOIN_PENDING[0] = False
    if not _AUX or not torch.cuda.is_available():
        return
    cur = torch.cuda.current_stream()
    for (idx, _), s in list(_AUX.items()):
        if idx == cur.device.index and s != cur:
            cur.wait_stream(s)


# Lazy pool gradients: behind the last Fire block of fire_blk1-3 sits SELayer + MaxPool (pointseg_net.py:27-46).  Its backward
# does not write the full-resolution gradient: it hands an UNINITIALISED tensor to autograd and leaves (pooled gradient, arg-max
# map, SE scale, SE-branch constant, pool row stride, stored) under the tensor's address.  The Fire block's cooperative
# BatchNorm backward routes the pooled gradient while it loads (dlio_bn_coop_bwd_pool); a bypass block passes the entry on with
# its squeeze data gradient (stored = True: that tensor then holds the rest of the sum) to the block in front of it.  Whoever
# cannot take an entry materialises it (lazy_materialize); an entry left over at the end of backward is an error.
_LAZY = {}
_LAZY_POOL = [os.environ.get("DLIO_LAZY_POOL_GRAD", "1") != "0"]
_LAZY_CB = [False]


class _LazyGrad:
    """what stands behind a gradient tensor that was handed to autograd unwritten: `tensor` (held, so that its address cannot
    be handed out again while the entry lives), the pooled gradient / arg-max map / SE scale / SE-branch constant / pool row
    stride it is formed from, `stored` (the tensor already holds a part of the sum), `token` (the identity of the Fire block
    whose backward is to consume it: FireFn.forward hangs the same object on its output), `full` (shape of the gradient)"""
    __slots__ = ("tensor", "dyp", "idx", "xs", "xadd", "sh", "stored", "token", "full")

    def __init__(self, tensor, dyp, idx, xs, xadd, sh, stored, token, full):
        self.tensor, self.dyp, self.idx, self.xs, self.xadd, self.sh = tensor, dyp, idx, xs, xadd, sh
        self.stored, self.token, self.full = stored, token, tuple(full)

    def pool(self):
        return (self.dyp, self.idx, self.xs, self.xadd, self.sh)


def lazy_clear():
    """forget every pending lazy gradient (start of a training step, error paths)"""
    _LAZY.clear()
    _LAZY_CB[0] = False


def _lazy_end_of_pass():
    _LAZY_CB[0] = False
    if _LAZY:
        n = len(_LAZY)
        _LAZY.clear()
        raise RuntimeError("%d lazy pool gradient(s) were not consumed by the Fire block they were meant for (functional._LAZY): "
                           "the backward pass did not reach it, or its gradient was taken directly -- DLIO_LAZY_POOL_GRAD=0 "
                           "turns the scheme off" % n)


def _lazy_put(entry):
    """register an unwritten gradient; the pass that created it checks at its end that somebody consumed it"""
    _LAZY[entry.tensor.data_ptr()] = entry
    if not _LAZY_CB[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_lazy_end_of_pass)
            _LAZY_CB[0] = True
        except RuntimeError:            # not inside an engine pass (a direct call in a test)
            pass


def _lazy_take(dout, token):
    """the entry behind dout if it is the one addressed to `token`; raises when the gradient of a block that expects a lazy
    entry arrives altered (a second consumer of the block's output, a tensor hook, retain_grad: autograd then sums the
    unwritten tensor into a new one)"""
    e = _LAZY.get(dout.data_ptr())
    if e is not None and (e.token is not token or token is None or tuple(e.tensor.shape) != tuple(dout.shape)):
        e = None
    if e is None and token is not None and any(v.token is token for v in _LAZY.values()):
        lazy_clear()
        raise RuntimeError("the lazy pool gradient addressed to this Fire block arrived altered: its output has a second "
                           "consumer, a tensor hook or retain_grad -- DLIO_LAZY_POOL_GRAD=0 turns the scheme off")
    return e


def lazy_materialize(t, entry=None):
    """t: a gradient tensor that may be lazy -> a tensor that holds the complete gradient (t itself when it has the
    gradient's shape; a new one when t stands for a gradient formed entirely from pooled tensors)"""
    e = entry if entry is not None else _LAZY.get(t.data_ptr())
    if e is None:
        return t
    _LAZY.pop(e.tensor.data_ptr(), None)
    same = tuple(t.shape) == e.full
    if e.stored:
        full = ops.maxpool2d_bwd(e.dyp, e.idx, e.full, 3, e.sh, 2, 1, 1, x_scale=e.xs, x_add=e.xadd)
        ops.ew_binary(t, full, 0, out=t)
        return t
    return ops.maxpool2d_bwd(e.dyp, e.idx, e.full, 3, e.sh, 2, 1, 1, x_scale=e.xs, x_add=e.xadd, out=t if same else None)


def _mark_lazy(ctx, out, ok):
    """FireFn.forward: `out` may receive a lazy gradient (the block's cooperative BatchNorm backward routes it)"""
    # a backward pass that raised never ran its end-of-pass callback: the flag it left set would keep every later pass from
    # registering the unconsumed-entry check again -- a forward pass is the one place no engine pass of this graph is running
    _LAZY_CB[0] = False
    tok = object() if ok else None
    ctx.lazy_token = tok
    out._dlio_lazy_token = tok


_NO_JOIN = [False]      # set while a hipGraph records a backward pass (a captured stream must not wait for outside work)


def _want_join():
    """called from backward nodes: join the auxiliary streams once when this backward pass ends
    (engine final callbacks run on the stream that called backward())"""
    if _AUX and not _JOIN_PENDING[0] and not _NO_JOIN[0]:
        _JOIN_PENDING[0] = True
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_aux_streams)
        except RuntimeError:        # not inside a backward pass (direct call in a test)
            _JOIN_PENDING[0] = False


class DeferredBranchFn(Function):
    """Re-attaches a branch that was computed early on its own stream at the point where its output
    is consumed.  The autograd engine runs ready nodes newest-first, so a branch issued first in
    forward (the IMU net: nets.py DeepLIO.forward) has its backward nodes run -- and their kernels
    enqueued behind an event on the main stream -- after everything issued later, i.e. after the
    whole encoder backward.  This node is created late, becomes ready as soon as the consumer's
    backward has produced the branch gradient, and runs the branch's backward right there on the
    branch's stream, where it overlaps the encoder backward instead of trailing it."""

    _ANCHOR = {}

    @staticmethod
    def attach(inner, stream):
        """inner: the branch output (still attached to the branch's own tape)"""
        key = inner.device
        a = DeferredBranchFn._ANCHOR.get(key)
        if a is None:       # the only differentiable input: makes the output require grad
            a = DeferredBranchFn._ANCHOR[key] = torch.zeros((), device=inner.device, requires_grad=True)
        return DeferredBranchFn.apply(a, (inner,), stream)

    @staticmethod
    def forward(ctx, anchor, inner, stream):
        # `inner` travels in a tuple: as a tensor argument its producer would become a child of
        # this node and be run a second time by the outer pass
        ctx.inner, ctx.stream = inner[0], stream
        return inner[0].detach()

    @staticmethod
    def backward(ctx, g):
        inner, s = ctx.inner, ctx.stream
        ctx.inner = None
        if DeferredBranchFn.LATE:
            # DLIO_DEFER_IMU_BWD=2 (off by default): issue the branch's backward when the engine has
            # issued everything else of this pass; the branch stream only waits for the point where
            # its gradient was produced (the event below).  Idea: the host holds back the encoder
            # backward's launches for ~1.2 ms while it issues the branch here (gpurun r02_c step dump).
            # Measured: 30.3 vs 28.8 ms/step (fp32), 37.6 vs 34.0 (bf16) -- the host is not far enough
            # ahead of the GPU at the end of the pass, the branch's latency-bound chain ends up exposed.
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            g.record_stream(s)
            DeferredBranchFn._PENDING.append((inner, g, s, ev))
            if len(DeferredBranchFn._PENDING) == 1:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(DeferredBranchFn.flush)
                except RuntimeError:            # not inside an engine pass
                    DeferredBranchFn.flush()
            return None, None, None
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            torch.autograd.backward([inner], [g])
        g.record_stream(s)
        _want_join()
        return None, None, None

    LATE = False                # (round 5 re-measured: 19.8 against 19.2 ms per step with the branch issued last)
    _PENDING = []

    @staticmethod
    def flush():
        pending, DeferredBranchFn._PENDING = DeferredBranchFn._PENDING, []
        for inner, g, s, ev in pending:
            s.wait_event(ev)
            with torch.cuda.stream(s):
                torch.autograd.backward([inner], [g])
        join_aux_streams()


def set_wgrad_stream(on):
    _WGRAD_FORK[0] = bool(on)


def _wgrad_stream(like):
    """companion stream of the current one for weight-gradient kernels (None = run inline)"""
    if not (_WGRAD_FORK[0] and like.is_cuda):
        return None
    return aux_stream(like.device, "wgrad@%x" % ops.raw_stream())


_PREP_SIDE = [os.environ.get("DLIO_PREP_SIDE", "0") != "0"]      # measured: median 18.63 ms with it, 18.38 without (ten alternations, round 6): off


def _prep_side_stream(cur):
    """where ops._PrepCache issues the long weight re-layout launches at the head of a step: the calling stream's weight-gradient
    companion (idle until backward) -- the stems then queue behind the short three-piece launches only"""
    if not (_PREP_SIDE[0] and _WGRAD_FORK[0]) or torch.cuda.is_current_stream_capturing():
        return None
    idx = cur.device.index if cur.device.index is not None else torch.cuda.current_device()
    return _AUX.get((idx, "wgrad@%x" % cur.cuda_stream)) or aux_stream(cur.device, "wgrad@%x" % cur.cuda_stream)


ops._PREP.side = _prep_side_stream


def wgrad_stream_of(cur):
    """the weight-gradient companion of stream `cur`, if one exists (None: nothing was forked from it)"""
    idx = cur.device.index if cur.device.index is not None else torch.cuda.current_device()
    return _AUX.get((idx, "wgrad@%x" % cur.cuda_stream))


def join_wgrad_stream():
    """current stream waits for the weight-gradient kernels forked from it so far"""
    if torch.cuda.is_available():
        cur = torch.cuda.current_stream()
        ws = _AUX.get((cur.device.index, "wgrad@%x" % cur.cuda_stream))
        if ws is not None:
            cur.wait_stream(ws)


_RNN_DIR_FORK = [True]


def _rnn_dir_stream(like):
    """companion stream for the reverse direction of a bidirectional RNN layer (None = run inline): the
    two directions are independent, latency-bound chains (a persistent kernel of 2 workgroups per
    direction), so running them side by side halves the branch's critical path"""
    if not (_RNN_DIR_FORK[0] and like.is_cuda):
        return None
    if torch.cuda.is_current_stream_capturing():     # hipGraph capture (tester.TestStep.capture): one stream per branch
        return None
    return aux_stream(like.device, "rnndir@%x" % ops.raw_stream())


def _forked(ws, fn, *tensors):
    """run fn() (a launch whose only output is a sunk gradient) on the companion stream ws"""
    ws.wait_stream(current_stream_obj(ws.device_index))
    with on_stream(ws):
        fn()
    for t in tensors:
        t.record_stream(ws)


def _sink(param, shape, like):
    """-> (out, accumulate, value_for_autograd)"""
    g = param.grad if (_SINK[0] and param is not None) else None
    if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == tuple(shape)
            and g.device == like.device):
        if like.is_cuda:
            _want_join()
        return g, True, None
    t = _new(tuple(shape), like)
    return t, False, t


# =============================================================================== conv + BN + ReLU
class _CBR:
    """conv (+bias) -> [ReLU] -> BatchNorm -> [ReLU] over channel slices, forward and backward.
    pre_relu  = conv->ReLU->BN  (FeatureNetSimple1, lidar_feat_nets.py:308-309)
    post_relu = conv->BN->ReLU  (everything else)."""

    @staticmethod
    def forward(x, x_ctot, x_coff, Cin, H, W, weight, bias, gamma, beta, rmean, rvar, stride, pad,
                training, momentum, eps, pre_relu, post_relu, raw, raw_ctot, raw_coff, out, out_ctot,
                out_coff, N, residual=None, r_ctot=0, r_coff=0, gap=None, gap_ctot=0, gap_coff=0,
                need_dx=True, in_aff=None, r_aff=None, stats_into=None, shift_into=None, conv_done=False,
                split_into=None, skip_bn=False, x_amax=None, want_amax=False):
        """in_aff (mean, scale, shift rows over the x_ctot input channels): x is stored BEFORE its producer's
        BatchNorm + ReLU and activated while the convolution loads it; r_aff: the same for the residual;
        stats_into (three [Cout] tensors): train-mode statistics only -- the activated output is not
        written, the consumers apply (mean, scale, beta) on load (apply-on-load, HISTORY 11).
        x_amax (one-float tensor: the largest |x|, left by x's producer): a 3x3 stride-1 layer runs its forward -- and in
        backward its weight gradient -- on the two-piece fp16 split; want_amax: the BatchNorm launch that writes `out` leaves
        its largest magnitude in d.out_amax (for the next layer)."""
        Cout, _, KH, KW = weight.shape
        d = ops.conv_desc(N, Cin, H, W, Cout, KH, KW, stride[0], stride[1], pad[0], pad[1],
                          in_ctot=x_ctot, in_coff=x_coff, out_ctot=raw_ctot, out_coff=raw_coff,
                          in_relu=1 if in_aff is not None else 0)
        bx3 = _use_bx3(N, Cin, Cout, KH, KW, stride, d.OH, d.OW)
        bx3 = bx3 and (KH == 3 or (pad[0] == 0 and pad[1] == 0))
        # the PointSeg stem (3x5, stride (1, 2), pointseg_net.py:18-20), forward only: MFMA-bound on the fp32 matrix
        # cores (61 % busy), half the MFMA time on the split-bf16 kernel
        stem = (not bx3 and _CONV_BX3[0] and _CONV_BX3_STEM[0] and in_aff is None
                and (((KH, KW) == (3, 5) and tuple(stride) == (1, 2) and (not need_dx or _CONV_BX3_3X5[0]))
                     or ((KH, KW) == (3, 3) and tuple(stride) == (2, 2) and _CONV_BX3_3X5[0] and Cin >= 16)))
        bx3 = bx3 or stem
        # FlowNet / ResNet / any plain conv + BN layer: 3x3 stride-1 forward on two fp16 pieces when the producer of x left its
        # largest magnitude (three MFMAs per product instead of six)
        h2f = (bx3 and not stem and KH == 3 and training and x_amax is not None and in_aff is None and not conv_done
               and _CONV_H2_FWD[0] and ops._SYNC_BN[0] is None and x.is_cuda and ops.conv3x3_h2_ok(d))
        # ... and the strided layers the split kernel takes (FlowNet conv2-6, ResNet's stage heads): the same with two pieces
        h2s = (stem and training and x_amax is not None and _CONV_H2_FWD[0] and ops._SYNC_BN[0] is None and x.is_cuda
               and Cout > 32 and not conv_done)
        d.x_amax = x_amax if h2f else None
        if h2f or h2s:
            wt = ops.conv_h2_prepped(weight, 0)
        elif bx3:
            wt = ops.conv_bx3_prepped(weight, 0)
        else:
            wt = ops.conv2d_prepped(weight, 0)
        if (training and need_dx and _use_bx3(N, Cout, Cin, KH, KW, stride, H, W)
                and (KH == 3 or (pad[0] == 0 and pad[1] == 0))):
            if (KH == 1 and _DGRAD1_H2[0] and ops._SYNC_BN[0] is None and not pre_relu and _BN_SMALL[0]
                    and (ops.bn_coop_ok(N, d.OH * d.OW) or (_SMALL_H2[0] and ops.bn_small_ok(N, d.OH * d.OW)))):
                # (with a cooperative geometry below _BN_COOP_MIN_BYTES the two-launch backward runs: it leaves amax as well)
                # 1x1 data gradients behind a one-launch BatchNorm backward (which leaves the largest |dy|): two fp16 pieces
                d.wh2_1 = ops.conv_h2_prepped(weight, 1)
            if (KH == 3 and _DGRAD_H2[0] and tuple(stride) == (1, 1) and ops._SYNC_BN[0] is None and not pre_relu and _BN_SMALL[0]
                    and (ops.bn_coop_ok(N, d.OH * d.OW) or (_SMALL_H2[0] and ops.bn_small_ok(N, d.OH * d.OW))
                         or (_CONV_H2_FWD[0] and want_amax))):
                # (want_amax: a plain conv + BN layer -- its two-launch BatchNorm backward leaves the largest |dy| as well)
                # ... and as two fp16 pieces where the data gradient runs on the two-piece kernel (it needs the largest
                # magnitude of its operand: conv_dgrad's `amax`, else the three-piece layout above is used)
                g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, 3, 3, 1, 1, 2 - d.PH, 2 - d.PW, OH=H, OW=W)
                if ops.conv3x3_h2_ok(g):
                    d.wh2_1 = ops.conv_h2_prepped(weight, 1)
            # the three-piece layout of the data-gradient direction (roles swapped) only where no two-piece one exists: every
            # registered layout is rebuilt at the head of each step, in front of both encoders (conv_dgrad fetches it on demand
            # should the producer of dy not have left a magnitude)
            d.bx3_1_ok = True
            if getattr(d, "wh2_1", None) is None:
                d.wbx3_1 = ops.conv_bx3_prepped(weight, 1)
        if training:
            # data-gradient layouts for backward: fetched here, where `weight` is the long-lived
            # Parameter (the cache identifies weights by object; backward only sees unpacked copies)
            plan = None
            if (d.SH > 1 or d.SW > 1) and need_dx and _DGRAD_PHASES[0]:
                plan = _phase_plan(d)
            # the fp32 layout only where conv_dgrad will run the fp32-MFMA kernel on it (every registered layout is
            # rebuilt once per optimizer step: the split-bf16 layers' fp32 copies were a third of that launch)
            if need_dx and getattr(d, "wbx3_1", None) is None and not getattr(d, "bx3_1_ok", False) and plan is None:
                d.wt2 = ops.conv2d_prepped(weight, 1)
            if plan is not None:               # tap-subset layouts of the phase-decomposed data gradient
                # (two fp16 pieces where the BatchNorm backward of this layer leaves the largest |dy|: any plain layer in
                #  training, see _CBR.backward)
                d.ph_h2 = (_CONV_H2_FWD[0] and want_amax and ops._SYNC_BN[0] is None and not pre_relu and Cin > 32)
                d.wt_ph = {(it[0], it[1]): ((ops.conv_h2_prepped_phase if d.ph_h2 else ops.conv_bx3_prepped_phase)(
                                                weight, d.SH, d.SW, it[0], it[1])
                                            if _phase_on_bx3(d, it[2], it[3]) else
                                            ops.conv2d_prepped_phase(weight, d.SH, d.SW, it[0], it[1]))
                           for it in plan if it is not None}
        if conv_done:
            pass                 # raw already holds this layer's output (fused Fire expand pair, dlio_fire_expand_fwd)
        elif h2f:
            ops.conv3x3_h2_fwd(x, x_amax, wt, bias, raw, d)
        elif h2s:
            ops.conv_h2_strided_fwd(x, x_amax, wt, bias, raw, d)
        elif stem:
            ops.conv3x5s2_bx3_fwd(x, wt, bias, raw, d)
        elif bx3 and KH == 1:
            ops.conv1x1_bx3_fwd(x, wt, bias, raw, d, in_aff=in_aff)
        elif bx3 and in_aff is None:
            ops.conv3x3_bx3_fwd(x, wt, bias, raw, d)
        elif bx3:
            # (the 3x3 split-bf16 kernel has no apply-on-load input: the fp32 kernel takes the transform)
            ops.conv2d_fwd(x, ops.conv2d_prepped(weight, 0), bias, raw, d, in_aff=in_aff)
        else:
            ops.conv2d_fwd(x, wt, bias, raw, d, in_aff=in_aff)
        OHW = d.OH * d.OW
        if skip_bn:                   # the caller runs this layer's BatchNorm itself (two layers in one launch, bn_small.hip)
            return d, None
        if split_into is not None:
            # the squeeze BatchNorm of a fused Fire block: activated tensor + its three-piece bf16 planes in one pass
            # (two-piece format: the bound on |out| its scale came from stays on the device for the expand3x3 weight gradient)
            d.out_bound = _new((1,), raw) if (split_into[1] and _WGRAD_H2[0]) else None
            prm = ops.bn_split16(raw, raw_ctot, raw_coff, gamma, beta, eps, momentum, rmean, rvar, out, out_ctot, out_coff,
                                 split_into[0], N, Cout, d.OH, d.OW, training, post_relu, fmt=split_into[1],
                                 bound_out=d.out_bound)
            return d, prm
        if stats_into is not None:
            prm = ops.bn_train_stats(raw, N, raw_ctot, raw_coff, Cout, OHW, pre_relu, gamma, eps, momentum, rmean, rvar,
                                     prm=stats_into, beta=beta, shift_out=shift_into)
            return d, prm
        if (training and _BN_SMALL[0] and _BN_COOP_FWD[0] and not pre_relu and raw.is_cuda and out is not None and ops.bn_coop_ok(N, OHW)
                and (gap is None or ops.bn_coop_gap_ok(N, OHW)) and _coop_pays(N, Cout, OHW)):
            # large planes: the N workgroups holding a channel in registers exchange their partial sums (one launch, one read)
            prm = torch.empty(3, Cout, dtype=torch.float32, device=raw.device)
            d.out_amax = ops.amax_slot_kept(raw.device) if want_amax else None
            ops.bn_coop_fwd(raw, raw_ctot, raw_coff, N, Cout, Cout, OHW, (gamma, beta, rmean, rvar), None, eps, momentum,
                            prm, out, out_ctot, out_coff, post_relu, residual=residual, r_ctot=r_ctot, r_coff=r_coff,
                            r_aff=r_aff, gap_out=gap, gap_ctot=gap_ctot, gap_coff=gap_coff, amax_out=d.out_amax)
            return d, prm
        if (training and _BN_SMALL[0] and not pre_relu and raw.is_cuda and ops.bn_small_ok(N, OHW) and stats_into is None):
            # small feature maps: statistics + apply in ONE launch, the tensor read once (bn_small.hip)
            prm = torch.empty(3, Cout, dtype=torch.float32, device=raw.device)
            d.out_amax = ops.amax_slot_kept(raw.device) if (want_amax and out is not None) else None
            ops.bn_small_fwd(raw, raw_ctot, raw_coff, N, Cout, Cout, OHW, (gamma, beta, rmean, rvar), None, eps, momentum,
                             prm, out, out_ctot, out_coff, post_relu, residual=residual, r_ctot=r_ctot, r_coff=r_coff,
                             r_aff=r_aff, gap_out=gap, gap_ctot=gap_ctot, gap_coff=gap_coff, amax_out=d.out_amax)
            return d, prm
        if training and _PLANE_BN[0]:
            # statistics + finalise + apply (+ the plane averages an SELayer wants) in 2 launches
            d.out_amax = ops.amax_slot_kept(raw.device) if (want_amax and raw.is_cuda) else None
            prm = ops.bn_train_apply(raw, raw_ctot, raw_coff, gamma, beta, eps, momentum, rmean, rvar, out,
                                     out_ctot, out_coff, N, Cout, OHW, pre_relu, post_relu, residual, r_ctot,
                                     r_coff, gap, gap_ctot, gap_coff, r_aff=r_aff, amax_out=d.out_amax)
            return d, prm
        if r_aff is not None:
            raise RuntimeError("a residual that is stored before its BatchNorm + ReLU (apply-on-load) needs the plane-"
                               "structured BatchNorm path (training with DLIO_PLANE_BN=1)")
        if training:
            prm = ops.bn_train_stats(raw, N, raw_ctot, raw_coff, Cout, OHW, pre_relu, gamma, eps,
                                     momentum, rmean, rvar)
        else:
            prm = ops.bn_eval_params(rmean, rvar, gamma, eps)
        ops.bn_apply(raw, raw_ctot, raw_coff, prm, beta, out, out_ctot, out_coff, N, Cout, OHW,
                     pre_relu, post_relu, residual, r_ctot, r_coff)
        return d, prm

    @staticmethod
    def backward(dy, dy_ctot, dy_coff, x, d, weight, bias, gamma, prm, beta, raw, training, pre_relu,
                 post_relu, draw, need_dx, dx=None, dx_ctot=0, dx_coff=0, dx_residual=None,
                 dxr_ctot=0, dxr_coff=0, dx_accumulate=False, in_aff=None, pooled=None, bn_grads=None, amax=None,
                 x_bound=None):
        """dy: grad wrt the activated output (slice).  draw: scratch [N,Cout,OH,OW] (contiguous).
        amax: one-float tensor with the largest |draw| (from the BatchNorm backward that wrote it), x_bound: one with a bound
        on |x| -- with both, the 3x3 weight gradient and data gradient run on the two-piece fp16 split.
        Returns (dweight, dbias, dgamma, dbeta); writes dx (slice) if need_dx:
        dx = dgrad (+ dx_residual) (+ previous dx when dx_accumulate)."""
        N, Cout, OHW = d.N, d.Cout, d.OH * d.OW
        if bn_grads is not None:     # draw already holds the BatchNorm's data gradient (bn_small.hip: two layers in one launch)
            ret_g, ret_b = bn_grads
        else:
            dgamma, acc_g, ret_g = _sink(gamma, (Cout,), dy)
            dbeta, acc_b, ret_b = _sink(beta, (Cout,), dy)
            if acc_g != acc_b:           # one flag serves both outputs of the reduce epilogue
                dgamma, acc_g, ret_g = _new((Cout,), dy), False, None
                dbeta, acc_b, ret_b = _new((Cout,), dy), False, None
                ret_g, ret_b = dgamma, dbeta
        if bn_grads is not None:
            pass
        elif (training and _BN_SMALL[0] and not pre_relu and dy.is_cuda and ops.bn_coop_ok(N, OHW) and pooled is None
              and _coop_pays(N, Cout, OHW)):
            if amax is None and ((need_dx and (getattr(d, "wh2_1", None) is not None or getattr(d, "ph_h2", False)))
                                 or getattr(d, "x_amax", None) is not None):
                amax = ops.amax_slot(dy.device)          # the data / weight gradient runs on two fp16 pieces: it wants the largest |draw|
            ops.bn_coop_bwd(dy, dy_ctot, dy_coff, raw, d.out_ctot, d.out_coff, prm, beta, None, draw, None, dgamma, dbeta,
                            None, None, acc_g, N, Cout, Cout, OHW, post_relu, amax_out=amax)
        elif (training and _BN_SMALL[0] and not pre_relu and dy.is_cuda and ops.bn_small_ok(N, OHW) and pooled is None):
            if amax is None and _SMALL_H2[0] and ((need_dx and (getattr(d, "wh2_1", None) is not None or getattr(d, "ph_h2", False)))
                                                  or getattr(d, "x_amax", None) is not None):
                amax = ops.amax_slot(dy.device)
            ops.bn_small_bwd(dy, dy_ctot, dy_coff, raw, d.out_ctot, d.out_coff, prm, beta, None, draw, None, dgamma, dbeta,
                             None, None, acc_g, N, Cout, Cout, OHW, post_relu, amax_out=amax)
        elif pooled is not None:
            # dy is the POOLED gradient: (arg-max map, pool row stride) -- the BatchNorm passes gather the gradient of the
            # activated tensor themselves (dlio_bn_bwd_pool), the pool's backward pass is not run
            ops.bn_bwd_pool(dy, pooled[0], raw, prm, beta, draw, pooled[1], dgamma, dbeta, accumulate=acc_g)
        elif _PLANE_BN[0]:
            if amax is None and dy.is_cuda and training and (getattr(d, "wh2_1", None) is not None or getattr(d, "x_amax", None) is not None
                                                              or getattr(d, "ph_h2", False)):
                amax = ops.amax_slot(dy.device)          # the data / weight gradient runs on two fp16 pieces
            ops.bn_bwd_fused(dy, dy_ctot, dy_coff, raw, d.out_ctot, d.out_coff, prm, beta, draw, Cout, 0, N,
                             Cout, OHW, pre_relu, post_relu, training, dgamma, dbeta, accumulate=acc_g, amax_out=amax)
        else:
            ops.bn_bwd(dy, dy_ctot, dy_coff, raw, d.out_ctot, d.out_coff, prm, beta, draw, Cout, 0, N,
                       Cout, OHW, pre_relu, post_relu, training, dgamma, dbeta, accumulate=acc_g)
        ret_bias = None
        if bias is not None:
            dbias, acc, ret_bias = _sink(bias, (Cout,), dy)
            if training and not pre_relu:
                # bias in front of a train-mode BN: d/db = sum(draw) = -scale*mean(g*xh)*sum(xh) and
                # sum(xh) = 0 identically -- the reference computes rounding noise (~1e-7) here
                if not acc:
                    dbias.zero_()
            else:
                ops.chan_sum(draw, N, Cout, 0, Cout, OHW, out=dbias, accumulate=acc)
        dw, acc_w, ret_w = _sink(weight, weight.shape, dy)
        dd = ops.conv_desc(N, d.Cin, d.H, d.W, Cout, d.KH, d.KW, d.SH, d.SW, d.PH, d.PW, OH=d.OH,
                           OW=d.OW, in_ctot=d.in_ctot, in_coff=d.in_coff, out_ctot=Cout, out_coff=0,
                           in_relu=1 if in_aff is not None else 0)
        ws = _wgrad_stream(dy) if acc_w else None
        if x_bound is None:
            x_bound = getattr(d, "x_amax", None)          # (a two-piece forward: the weight gradient takes the same scale)
            if x_bound is not None and not ops.amax_fresh(x_bound):
                x_bound = None                            # (recycled since the forward pass: the three-piece weight gradient)
        wg_h2 = (amax is not None and x_bound is not None and in_aff is None and _WGRAD_H2[0] and d.KH == 3
                 and ops.conv3x3_wgrad_h2_ok(dd))
        if wg_h2 and ws is None:
            ops.conv3x3_wgrad_h2(x, x_bound, draw, amax, dw, dd, accumulate=acc_w)
        elif wg_h2:
            _forked(ws, lambda: ops.conv3x3_wgrad_h2(x, x_bound, draw, amax, dw, dd, accumulate=True), draw, x, amax, x_bound)
        elif ws is None:
            ops.conv2d_wgrad(x, draw, dw, dd, in_aff=in_aff, accumulate=acc_w)
        else:
            # the weight gradient goes straight into the flat gradient buffer and nothing on the
            # tape waits for it: fork it so the data-gradient chain continues immediately
            # (every operand is marked as in use on the companion stream: the apply-on-load constants too -- they are
            #  the last thing a deferred Fire block frees, and a reused block under a queued kernel gave wrong
            #  gradients for the first layers of backward, state-dependently)
            _forked(ws, lambda: ops.conv2d_wgrad(x, draw, dw, dd, in_aff=in_aff, accumulate=True), draw, x,
                    *(in_aff if in_aff is not None else ()))
        if need_dx:
            conv_dgrad(draw, weight, d, dx, dx_ctot, dx_coff, dx_residual, dxr_ctot, dxr_coff,
                       dx_accumulate, amax=amax)
        return ret_w, ret_bias, ret_g, ret_b


def conv_dgrad(dy, weight, d, dx, dx_ctot, dx_coff, residual=None, r_ctot=0, r_coff=0,
               accumulate=False, amax=None):
    """dy: contiguous [N,Cout,OH,OW] -> dx channel slice.  Stride-1 convs reuse the MFMA
    forward kernel with tap-reversed, transposed weights.  amax: one-float tensor with max |dy| from dy's producer."""
    N, Cin, Cout = d.N, d.Cin, d.Cout
    if accumulate:
        residual, r_ctot, r_coff = dx, dx_ctot, dx_coff
    wb = getattr(d, "wbx3_1", None) if (d.SH == 1 and d.SW == 1) else None
    wh = getattr(d, "wh2_1", None) if amax is not None else None
    if wh is None and wb is None and getattr(d, "bx3_1_ok", False) and d.SH == 1 and d.SW == 1:
        wb = ops.conv_bx3_prepped(weight, 1)             # (a two-piece layer whose operand came without its magnitude)
    if wh is not None and d.KH == 3:
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, 3, 3, 1, 1, 2 - d.PH, 2 - d.PW, OH=d.H, OW=d.W, in_ctot=Cout,
                          in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff, res_ctot=r_ctot, res_coff=r_coff)
        ops.conv3x3_h2_fwd(dy, amax, wh, None, dx, g, residual=residual)
    elif wh is not None and d.KH == 1:
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, 1, 1, 1, 1, 0, 0, OH=d.H, OW=d.W, in_ctot=Cout,
                          in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff, res_ctot=r_ctot, res_coff=r_coff)
        ops.conv1x1_h2_fwd(dy, amax, wh, None, dx, g, residual=residual)
    elif wb is not None and d.KH == 1:
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, 1, 1, 1, 1, 0, 0, OH=d.H, OW=d.W, in_ctot=Cout,
                          in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff, res_ctot=r_ctot, res_coff=r_coff)
        ops.conv1x1_bx3_fwd(dy, wb, None, dx, g, residual=residual)
    elif wb is not None:
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, 3, 3, 1, 1, 2 - d.PH, 2 - d.PW, OH=d.H, OW=d.W, in_ctot=Cout,
                          in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff, res_ctot=r_ctot, res_coff=r_coff)
        ops.conv3x3_bx3_fwd(dy, wb, None, dx, g, residual=residual)
    elif d.SH == 1 and d.SW == 1:
        wt2 = getattr(d, "wt2", None)
        if wt2 is None:
            wt2 = ops.conv2d_prepped(weight, 1)
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, d.KH, d.KW, 1, 1, d.KH - 1 - d.PH,
                          d.KW - 1 - d.PW, OH=d.H, OW=d.W, in_ctot=Cout, in_coff=0, out_ctot=dx_ctot,
                          out_coff=dx_coff, res_ctot=r_ctot, res_coff=r_coff)
        ops.conv2d_fwd(dy, wt2, None, dx, g, residual=residual)
    else:
        # strided conv: insert the stride's zeros into dy and run the stride-1 MFMA kernel on it
        # (SH*SW x the minimal MFMA work, still ~50x faster than a scalar gather)
        if _DGRAD_PHASES[0] and _dgrad_phases(dy, weight, d, dx, dx_ctot, dx_coff, residual, r_ctot, r_coff, amax=amax):
            return dx
        HU = (d.OH - 1) * d.SH + 1 + (d.H + 2 * d.PH - d.KH) % d.SH
        WU = (d.OW - 1) * d.SW + 1 + (d.W + 2 * d.PW - d.KW) % d.SW
        up = ops.zero_upsample2d(dy, HU, WU, d.SH, d.SW)
        wt2 = getattr(d, "wt2", None)
        if wt2 is None:
            wt2 = ops.conv2d_prepped(weight, 1)
        g = ops.conv_desc(N, Cout, HU, WU, Cin, d.KH, d.KW, 1, 1, d.KH - 1 - d.PH, d.KW - 1 - d.PW,
                          OH=d.H, OW=d.W, in_ctot=Cout, in_coff=0, out_ctot=dx_ctot, out_coff=dx_coff,
                          res_ctot=r_ctot, res_coff=r_coff)
        ops.conv2d_fwd(up, wt2, None, dx, g, residual=residual)
    return dx


_CONV_BX3 = [os.environ.get("DLIO_CONV_BX3", "1") != "0"]
_FIRE_FUSED = [os.environ.get("DLIO_FIRE_FUSED", "1") != "0"]
_DGRAD_H2 = [os.environ.get("DLIO_DGRAD_H2", "1") != "0"]       # 3x3 data gradients of fire_blk1-3 on two fp16 pieces
_FIRE_H2 = [os.environ.get("DLIO_FIRE_H2", "1") != "0"]         # fused Fire forward on two fp16 pieces (training)
_DGRAD1_H2 = [os.environ.get("DLIO_DGRAD1_H2", "1") != "0"]     # squeeze / expand1x1 data gradients of fire_blk1-3 likewise
_SMALL_H2 = [os.environ.get("DLIO_SMALL_H2", "1") != "0"]       # ... and of fire_blk4 / blk5 (scale from dlio_bn_small_bwd's amax_out)
_WGRAD_H2 = [os.environ.get("DLIO_WGRAD_H2", "1") != "0"]       # expand3x3 weight gradients of fire_blk1-3 on two fp16 pieces
# plain conv + BN layers (FlowNet, ResNet): 3x3 stride-1 forward + weight gradient on two fp16 pieces, scale from the producer's
# BatchNorm launch; data gradients likewise behind the two-launch BatchNorm backward (0: three-piece bf16 as in round 4)
_CONV_H2_FWD = [os.environ.get("DLIO_CONV_H2_FWD", "1") != "0"]
_FIRE_STATS = [os.environ.get("DLIO_FIRE_STATS", "1") != "0"]   # apply-on-load blocks: BatchNorm statistics from the expand launch
# ... and for the blocks on large planes that write their output: statistics from the expand launch + a streaming apply
# (0: the cooperative one-launch BatchNorm forward of round 4)
_FIRE_STREAM = [os.environ.get("DLIO_FIRE_STREAM", "1") != "0"]
# the block in front of SELayer + MaxPool pools while it applies its BatchNorm: its output is never written (0: written, pooled by SEPoolFn)
_POOL_FUSE = [os.environ.get("DLIO_POOL_FUSE", "1") != "0"]


# the cooperative one-launch BatchNorm kernels pay from this many bytes per operand on: below it (the squeeze layers: 8-34 MB) the
# second read of the two-launch kernels comes out of the L2 / Infinity Cache and they are faster than an exchange between
# workgroups (tools/bench_bn_squeeze.py: 26-31 us against 32-48)
# ... alone; inside the five-stream step the two choices are equal within the run-to-run noise (19.1-19.4 ms either way), so the
# default keeps the launch count down: 0 = cooperative wherever the geometry allows
_BN_COOP_MIN_BYTES = [0]            # (a module switch for tools / ablations: bytes per operand)


def _coop_pays(N, C, HW):
    return 4 * N * C * HW >= _BN_COOP_MIN_BYTES[0]


def _pool_row_stride(pool):
    """row stride of a (k, stride, pad) max-pool the fused kernels take (3 x 3, stride (1 | 2, 2), padding 1), else 0"""
    if pool is None:
        return 0
    k, stride, pad = pool
    k = k if isinstance(k, (tuple, list)) else (k, k)
    stride = stride if isinstance(stride, (tuple, list)) else (stride, stride)
    pad = pad if isinstance(pad, (tuple, list)) else (pad, pad)
    if tuple(k) == (3, 3) and tuple(pad) == (1, 1) and stride[1] == 2 and stride[0] in (1, 2):
        return int(stride[0])
    return 0
_PAIR_FUSE = [os.environ.get("DLIO_PAIR_FUSE", "1") != "0"]   # gap + add / sub + fc1 + act of the lidar head as one launch
_SE_FC = [os.environ.get("DLIO_SE_FC", "1") != "0"]         # the SELayer's fc pair as one launch (csrc/se_fc.hip)
_SE_POOLED_DOT = [os.environ.get("DLIO_SE_POOLED_DOT", "1") != "0"]   # SELayer scale gradient from pooled tensors
_BN_COOP_FWD = [True]          # cooperative kernels in the forward pass too (a module switch for ablations)
# both expand data gradients in one launch (dlio_fire_expand_dgrad): built, tested, OFF -- the expand1x1 chunks pay the full
# patch staging for a ninth of the MFMA work (isolated 169 vs 155 us at blk1, step 23.0 vs 22.7 ms: DESIGN 9)
_FIRE_FUSED_DGRAD = [os.environ.get("DLIO_FIRE_FUSED_DGRAD", "0") != "0"]
_BN_SMALL = [os.environ.get("DLIO_BN_SMALL", "1") != "0"]           # one-launch BatchNorm of small feature maps (bn_small.hip)      # expand1x1 || expand3x3 in one launch (fire_expand.hip)
_CONV_BX3_1X1 = [os.environ.get("DLIO_CONV_BX3_1X1", "1") != "0"]
_CONV_BX3_STEM = [True]
# FlowNet conv2 / conv3 (3x5, stride (1, 2), 64 / 128 input channels) and the 3x3 stride-2 layers (FlowNet conv4-6, ResNet)
_CONV_BX3_3X5 = [True]
_BX3_1X1_KSPLIT = [os.environ.get("DLIO_BX3_1X1_KSPLIT", "1") != "0"]
_BX3_1X1_KSPLIT_PIX = [16384]
_BX3_1X1_MIN = [int(v) for v in os.environ.get("DLIO_BX3_1X1_MIN", "16,16,65536,8192").split(",")]   # Cin, Cout, pixels, pixels (widening layers)


def set_conv_bx3(on):
    _CONV_BX3[0] = bool(on)


# Data-parallel PARITY runs (synchronised BatchNorm: a global batch split over ranks must train exactly like the same batch
# in one process) route the 1x1 layers by the GLOBAL launch size, so that every world size picks the same kernels; throughput
# runs (per-replica statistics) route by what each GPU actually launches.  Set by dist.GradSync.enable_sync_bn.
_ROUTE_WORLD = [1]


def _use_bx3(N, Cin, Cout, KH, KW, stride, OH, OW):
    """3x3 stride-1 convolutions go to the split-bf16 kernel (conv_bx3.hip: fp32-accurate products
    from six bf16 MFMAs, 1.3-1.6x the fp32-MFMA kernel on every PointSeg / FlowNet / ResNet shape,
    tools/bench_bx3.py)"""
    if not _CONV_BX3[0] or tuple(stride) != (1, 1):
        return False
    if (KH, KW) == (3, 3):
        return True
    # 1x1 (conv1x1_bx3_kernel, two waves per SIMD): every layer with enough pixels to fill the chip (blk1-3:
    # 28-90 us = 4.1-6 TB/s against 35-135 us on the fp32-MFMA float4 kernel, tools/conv1x1_table.py); of the small
    # ones (blk4/5) the widening layers (few input channels, >= 4x as many output channels: their time is the
    # store) -- the narrowing ones have a long K loop and few waves, the fp32 split-K kernel is faster there
    if (KH, KW) != (1, 1) or not _CONV_BX3_1X1[0] or (OH * OW) % 4 or Cin < _BX3_1X1_MIN[0] or Cout < _BX3_1X1_MIN[1]:
        return False
    pix = N * OH * OW * _ROUTE_WORLD[0]
    if pix >= _BX3_1X1_MIN[2]:
        return True
    # small layers: the widening ones, and on the smallest maps (blk5: <= 16 k pixels) the narrowing ones with >= 192
    # input channels, whose channel loop the kernel splits over workgroups (dlio_conv1x1_bx3_fwd_ws: 39-55 -> 22-25 us;
    # at 32 k pixels the slabs cost more than the fp32 split-K kernel's 30 us)
    return pix >= _BX3_1X1_MIN[3] and ((Cin <= 128 and Cout >= 4 * Cin) or
                                       (_BX3_1X1_KSPLIT[0] and Cin >= 192 and pix <= _BX3_1X1_KSPLIT_PIX[0]))


_DGRAD_PHASES = [os.environ.get("DLIO_DGRAD_PHASES", "1") != "0"]
_PHASE_KERNELS = {(3, 3), (3, 2), (3, 1), (2, 2), (2, 1), (1, 2), (1, 1), (3, 5), (5, 7)}   # stride-1 instantiations


def set_dgrad_phases(on, bx3=None, bx3_min_k=None):
    _DGRAD_PHASES[0] = bool(on)
    if bx3 is not None:
        _DGRAD_PHASES_BX3[0] = bool(bx3)
    if bx3_min_k is not None:
        _PHASE_BX3_MIN_K[0] = int(bx3_min_k)


# the phases' stride-1 convolutions on the split-bf16 kernel (dlio_conv_bx3_fwd_taps) instead of the fp32 MFMA
_DGRAD_PHASES_BX3 = [True]
_PHASE_BX3_KERNELS = {(3, 3), (3, 2), (2, 2), (2, 1), (1, 2), (1, 1)}
_PHASE_BX3_MIN_K = [16]      # channels of dy below which the 16-channel chunks of the split-bf16 kernel are mostly padding


def _phase_on_bx3(d, Mh, Mw):
    return _CONV_BX3[0] and _DGRAD_PHASES_BX3[0] and (Mh, Mw) in _PHASE_BX3_KERNELS and d.Cout >= _PHASE_BX3_MIN_K[0]


def _phase_plan(d):
    """per input phase (a, b) of a strided convolution: (rh, rw, Mh, Mw, pad_top, pad_left, Hp, Wp) of
    the stride-1 convolution of dy that produces dx[a::SH, b::SW]; None entries are phases without
    taps; returns None when a phase needs something the stride-1 kernels do not have.
    Rows ih = SH*i + a only see the taps kh = rh + SH*m with rh = (a + PH) mod SH, and
    dx[SH*i + a] = sum_m w[rh + SH*m] dy[i + qh - m], qh = (a + PH) div SH -- a correlation with the
    reversed tap subset under a top padding of Mh-1-qh (the bottom padding is implied by the
    output extent)."""
    SH, SW = d.SH, d.SW
    if SH * SW > 4:
        return None
    plan = []
    for a in range(SH):
        rh, qh = (a + d.PH) % SH, (a + d.PH) // SH
        Mh = len(range(rh, d.KH, SH))
        Hp = len(range(a, d.H, SH))
        for b in range(SW):
            rw, qw = (b + d.PW) % SW, (b + d.PW) // SW
            Mw = len(range(rw, d.KW, SW))
            Wp = len(range(b, d.W, SW))
            if Hp == 0 or Wp == 0:
                return None
            if Mh == 0 or Mw == 0:
                plan.append(None)
                continue
            pt, pl = Mh - 1 - qh, Mw - 1 - qw
            lo_h, lo_w = d.OH + 2 * pt - Mh + 1, d.OW + 2 * pl - Mw + 1
            if (pt < 0 or pl < 0 or (Mh, Mw) not in _PHASE_KERNELS or not (lo_h <= Hp <= lo_h + Mh - 1)
                    or not (lo_w <= Wp <= lo_w + Mw - 1)):
                return None
            plan.append((rh, rw, Mh, Mw, pt, pl, Hp, Wp))
    return plan


def _dgrad_phases(dy, weight, d, dx, dx_ctot, dx_coff, residual, r_ctot, r_coff, amax=None):
    """strided data gradient as SH*SW stride-1 convolutions of dy, one per input phase, woven
    together by dlio_phase_interleave2d: no MFMA work on inserted zeros (the zero-upsample route
    does SH*SW x the minimum).  Returns False when the plan does not exist."""
    plan = _phase_plan(d)
    if plan is None:
        return False
    N, Cin, Cout = d.N, d.Cin, d.Cout
    stash = getattr(d, "wt_ph", None)
    phases = []
    for item in plan:
        if item is None:
            phases.append(None)
            continue
        rh, rw, Mh, Mw, pt, pl, Hp, Wp = item
        bx3 = _phase_on_bx3(d, Mh, Mw)
        ph_h2 = bx3 and getattr(d, "ph_h2", False)          # (the stash then holds two-piece layouts)
        h2 = ph_h2 and amax is not None and stash is not None
        wt = stash.get((rh, rw)) if stash else None
        if wt is None or (ph_h2 and not h2):
            # direct call (no forward ran on this descriptor) / a two-piece layout whose operand came without its magnitude
            wt = (ops.conv_bx3_prepped_phase if bx3 else ops.conv2d_prepped_phase)(weight, d.SH, d.SW, rh, rw, cache=False)
        out = _new((N, Cin, Hp, Wp), dy)
        g = ops.conv_desc(N, Cout, d.OH, d.OW, Cin, Mh, Mw, 1, 1, pt, pl, OH=Hp, OW=Wp, in_ctot=Cout, in_coff=0,
                          out_ctot=Cin, out_coff=0)
        if h2:
            ops.conv_h2_taps_fwd(dy, amax, wt, None, out, g)
        elif bx3:
            ops.conv_bx3_taps_fwd(dy, wt, None, out, g)
        else:
            ops.conv2d_fwd(dy, wt, None, out, g)
        phases.append(out)
    ops.phase_interleave2d(phases, d.SH, d.SW, dx, dx_ctot, dx_coff, N, Cin, d.H, d.W, residual, r_ctot, r_coff)
    return True


class ConvBnAct(Function):
    """One conv+BN(+ReLU) layer.  Used by the PointSeg stem (pointseg_net.py:18-20), FlowNet
    conv() blocks (base_net.py:55-71), ResNet (resnet.py:36-38 + BasicBlock) and Simple-1."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, rmean, rvar, stride, pad, training, momentum,
                eps, pre_relu, post_relu):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        x_amax = getattr(x, "_dlio_amax", None)
        if x_amax is not None and not ops.amax_fresh(x_amax):
            x_amax = None                 # (the producer's slot was recycled since: the three-piece kernels need no scale)
        raw = _new((N, Cout, OH, OW), x)
        out = _new((N, Cout, OH, OW), x)
        d, prm = _CBR.forward(x, Cin, 0, Cin, H, W, weight, bias, gamma, beta, rmean, rvar, stride,
                              pad, training, momentum, eps, pre_relu, post_relu, raw, Cout, 0, out,
                              Cout, 0, N, need_dx=ctx.needs_input_grad[0], x_amax=x_amax,
                              want_amax=training and x.is_cuda and _CONV_H2_FWD[0] and ops._SYNC_BN[0] is None)
        ctx.save_for_backward(x, weight, beta, raw, prm, gamma, bias)
        ctx.cfg = (d, training, pre_relu, post_relu)
        out._dlio_amax = getattr(d, "out_amax", None)        # (the largest |out|: the next layer's two-piece operand scale)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, beta, raw, prm, gamma, bias = ctx.saved_tensors
        d, training, pre_relu, post_relu = ctx.cfg
        dy = dy.contiguous()
        draw = torch.empty_like(raw)
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        dw, db, dg, dbt = _CBR.backward(dy, d.Cout, 0, x, d, weight, bias, gamma, prm, beta, raw,
                                        training, pre_relu, post_relu, draw, need_dx, dx, d.Cin, 0)
        return (dx, dw, db, dg, dbt) + (None,) * 9


class ConvBnActPoolFn(Function):
    """conv -> BatchNorm -> ReLU -> MaxPool2d(3, pad 1, stride (1|2, 2)) as one tape node with the BatchNorm + ReLU applied
    by the pool while it loads (training): the activated tensor is never written.  The PointSeg stem + pool1
    (pointseg_net.py:18-21: 268 MB per encoder at the headline shape)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, rmean, rvar, stride, pad, momentum, eps, pk, pstride, ppad):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        OH = (H + 2 * pad[0] - KH) // stride[0] + 1
        OW = (W + 2 * pad[1] - KW) // stride[1] + 1
        raw = _new((N, Cout, OH, OW), x)
        aff, inv = _new((3, Cout), x), _new((Cout,), x)
        d, _ = _CBR.forward(x, Cin, 0, Cin, H, W, weight, bias, gamma, beta, rmean, rvar, stride, pad, True, momentum, eps,
                            False, True, raw, Cout, 0, None, Cout, 0, N, need_dx=ctx.needs_input_grad[0],
                            stats_into=(aff[0], inv, aff[1]), shift_into=aff[2])
        y, idx = ops.maxpool2d_fwd_aff(raw, aff, pk, pstride[0], pstride[1], ppad[0], ppad[1])
        ctx.save_for_backward(x, weight, beta, raw, aff, inv, gamma, bias, idx)
        ctx.cfg = (d, pk, pstride, ppad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, beta, raw, aff, inv, gamma, bias, idx = ctx.saved_tensors
        d, pk, pstride, ppad = ctx.cfg
        dy = dy.contiguous()
        draw = torch.empty_like(raw)
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        # the pool's backward folded into the BatchNorm backward (not with synchronised statistics: that path needs the
        # partials all-reduced between the two launches)
        fold = _POOL_BN_BWD[0] and ops._SYNC_BN[0] is None
        dact = dy if fold else ops.maxpool2d_bwd(dy, idx, tuple(raw.shape), pk, pstride[0], pstride[1], ppad[0], ppad[1])
        dw, db, dg, dbt = _CBR.backward(dact, d.Cout, 0, x, d, weight, bias, gamma, (aff[0], inv, aff[1]), beta, raw,
                                        True, False, True, draw, need_dx, dx, d.Cin, 0,
                                        pooled=(idx, pstride[0]) if fold else None)
        return (dx, dw, db, dg, dbt) + (None,) * 9


# =============================================================================== Fire
class FireFn(Function):
    """Fire block (pointseg_modules.py:116-142) as ONE tape node: squeeze CBR, then the two
    expand convolutions write the halves of the concatenated output; 'simple' bypass is the
    residual operand of the BN-apply kernels.

    Apply-on-load (training, blocks without bypass whose only consumer is the next Fire block):
    `defer` -- the expand BatchNorms only take their statistics; the block returns its RAW expand
    output and an [3, CE] table (mean, scale, beta) instead of the activated tensor, which is never
    written.  `x_aff` -- the input is such a pair: the squeeze convolution, the bypass residual and (in
    backward) the squeeze weight gradient apply max(0, (x - mean) * scale + beta) while they load.  The raw
    tensor stands in for the activated one on the tape: its gradient IS d loss / d activated output."""

    @staticmethod
    def forward(ctx, x, sw, sb, sg, sbe, srm, srv, e1w, e1b, e1g, e1be, e1rm, e1rv, e3w, e3b, e3g,
                e3be, e3rm, e3rv, training, momentum, eps, bypass, want_gap=False, x_aff=None, defer=False, pool=None):
        """pool (k, stride, pad): the block is followed by SELayer + MaxPool2d(pool) and NOTHING else reads its output
        (pointseg_net.py:27-46) -- where the streaming path applies the block returns (pooled maximum of its output, plane
        averages, arg-max map) instead of the output (see the `poolfuse` branch)"""
        ctx.set_materialize_grads(False)     # the by-products (plane averages, affine table) get no gradient: no zero fills
        ctx.x_token = getattr(x, "_dlio_lazy_token", None)      # (the producer of x takes a lazy gradient: see _LAZY)
        ctx.lazy_token = None
        ctx.poolfuse = None
        x = x.contiguous()
        N, Cin, H, W = x.shape
        S_, E1, E3 = sw.shape[0], e1w.shape[0], e3w.shape[0]
        CE = E1 + E3
        raw_s = _new((N, S_, H, W), x)
        act_s = _new((N, S_, H, W), x)
        # expand1x1 || expand3x3 as one launch on the squeeze activation's split planes (csrc/fire_expand.hip)
        fused = (_FIRE_FUSED[0] and x.is_cuda and E1 == E3 and W % 4 == 0 and _CONV_BX3[0]
                 and tuple(e3w.shape[2:]) == (3, 3) and tuple(e1w.shape[2:]) == (1, 1))
        planes = ops.fire_planes(N, S_, H, W, x.device) if fused else None
        # training: the planes and the expand weights as two fp16 pieces (three MFMAs per product instead of six)
        h2 = 1 if (fused and training and _FIRE_H2[0] and ops._SYNC_BN[0] is None) else 0
        d_s, prm_s = _CBR.forward(x, Cin, 0, Cin, H, W, sw, sb, sg, sbe, srm, srv, (1, 1), (0, 0),
                                  training, momentum, eps, False, True, raw_s, S_, 0, act_s, S_, 0, N,
                                  in_aff=x_aff, split_into=(planes, h2) if fused else None)
        raw_e = _new((N, CE, H, W), x)
        res = x if bypass else None
        # an apply-on-load block takes its BatchNorm statistics out of the expand launch's epilogue (tile sums + one small
        # finalising launch) instead of a pass over the concat buffer
        # ... and so does a block on LARGE planes that writes its output (bypass blocks, the block in front of SELayer + pool):
        # with the statistics known its BatchNorm + ReLU (+ residual) is a streaming apply -- round 4 ran the cooperative
        # one-launch kernel here, whose workgroups spin for each other's partial sums
        stream_bn = (fused and not defer and training and _FIRE_STATS[0] and _FIRE_STREAM[0] and ops._SYNC_BN[0] is None
                     and _BN_SMALL[0] and ops.bn_coop_ok(N, H * W))
        epi = fused and training and _FIRE_STATS[0] and ops._SYNC_BN[0] is None and (defer or stream_bn)
        if fused:
            w3p, w1p = ((ops.conv_h2_prepped(e3w), ops.conv_h2_prepped(e1w)) if h2
                        else (ops.conv_bx3_prepped(e3w, 0), ops.conv_bx3_prepped(e1w, 0)))
        if epi:
            aff, inv = _new((3, CE), x), _new((CE,), x)
            ops.fire_expand_fwd_stats(planes, w3p, w1p, e3b, e1b, raw_e, N, S_, H, W, E1, CE, 0, (e1g, e1be, e1rm, e1rv),
                                      (e3g, e3be, e3rm, e3rv), eps, momentum, aff[0], inv, aff[1], aff[2], fmt=h2)
            del planes
        elif fused:
            ops.fire_expand_fwd(planes, w3p, w1p, e3b, e1b, raw_e, N, S_, H, W, E1, CE, 0, fmt=h2)
            del planes
        # both expand data gradients in one launch (backward): the layouts are fetched here, where the weights are the
        # long-lived Parameters the cache knows
        wdg = ((ops.conv_bx3_prepped(e3w, 1), ops.conv_bx3_prepped(e1w, 1))
               if (training and _FIRE_FUSED_DGRAD[0] and _CONV_BX3[0] and x.is_cuda and tuple(e3w.shape[2:]) == (3, 3)
                   and tuple(e1w.shape[2:]) == (1, 1)) else None)
        ctx.wdg = wdg
        lazy_ok = (_LAZY_POOL[0] and training and x.is_cuda and _BN_SMALL[0] and ops._SYNC_BN[0] is None and wdg is None
                   and ops.bn_coop_ok(N, H * W))
        coop = (training and _BN_SMALL[0] and _BN_COOP_FWD[0] and x.is_cuda and not defer and ops.bn_coop_ok(N, H * W)
                and (not want_gap or ops.bn_coop_gap_ok(N, H * W)))
        small = training and _BN_SMALL[0] and x.is_cuda and (coop or ops.bn_small_ok(N, H * W))
        bn_fwd = ops.bn_coop_fwd if coop else ops.bn_small_fwd
        if epi:
            d_1, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e1w, e1b, e1g, e1be, e1rm, e1rv, (1, 1), (0, 0), training,
                                  momentum, eps, False, True, raw_e, CE, 0, None, CE, 0, N, conv_done=True, skip_bn=True)
            d_3, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e3w, e3b, e3g, e3be, e3rm, e3rv, (1, 1), (1, 1), training,
                                  momentum, eps, False, True, raw_e, CE, E1, None, CE, E1, N, conv_done=True, skip_bn=True)
            ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, aff, inv[:E1], inv[E1:],
                                  sb, sg, e1b, e1g, e3b, e3g, x_aff)
            ctx.cfg = (d_s, d_1, d_3, training, bypass, True)
            ctx.small_prm = (aff[0], inv, aff[1])
            if defer:
                ctx.mark_non_differentiable(aff)
                _mark_lazy(ctx, raw_e, lazy_ok)
                return raw_e, aff
            raff = x_aff if bypass else None
            sh = _pool_row_stride(pool)
            if (want_gap and sh and lazy_ok and _POOL_FUSE[0] and ops.bn_coop_pool_ok(N, H, W, sh)
                    and ops.bn_aff_pool_ok(H, W, sh)):
                # poolfuse: BatchNorm + ReLU + bypass + the max-pool behind the SELayer in one pass; the block's output is
                # never written (its gradient arrives as a lazy entry formed from the POOLED gradient: SEPoolFn.backward)
                pr, idx, gap = ops.bn_aff_pool_fwd(raw_e, CE, 0, N, CE, H, W, sh, aff, residual=res, r_ctot=Cin, r_coff=0,
                                                   r_aff=raff)
                ctx.poolfuse = (sh, (N, CE, H, W))
                ctx.mark_non_differentiable(gap, idx)
                _mark_lazy(ctx, pr, True)
                pr._dlio_pooled = (idx, H, W)
                return pr, gap, idx
            out = _new((N, CE, H, W), x)
            gap = _new((N, CE), x) if want_gap else None
            ops.bn_aff_apply(raw_e, CE, 0, N, CE, H * W, aff, out, CE, 0, residual=res, r_ctot=Cin, r_coff=0, r_aff=raff,
                             gap_out=gap, gap_ctot=CE, gap_coff=0)
            _mark_lazy(ctx, out, lazy_ok)
            if not want_gap:
                return out
            ctx.mark_non_differentiable(gap)
            return out, gap
        if small:
            # fire_blk4 / fire_blk5: the two expand BatchNorms as ONE launch that reads the concat buffer once
            d_1, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e1w, e1b, e1g, e1be, e1rm, e1rv, (1, 1), (0, 0), training,
                                  momentum, eps, False, True, raw_e, CE, 0, None, CE, 0, N, conv_done=fused, skip_bn=True)
            d_3, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e3w, e3b, e3g, e3be, e3rm, e3rv, (1, 1), (1, 1), training,
                                  momentum, eps, False, True, raw_e, CE, E1, None, CE, E1, N, conv_done=fused, skip_bn=True)
            sets = ((e1g, e1be, e1rm, e1rv), (e3g, e3be, e3rm, e3rv))
            if defer:
                aff, inv = _new((3, CE), x), _new((CE,), x)
                ops.bn_small_fwd(raw_e, CE, 0, N, CE, E1, H * W, sets[0], sets[1], eps, momentum, (aff[0], inv, aff[1]),
                                 None, CE, 0, True, shift_out=aff[2])
                ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, aff, inv[:E1], inv[E1:],
                                      sb, sg, e1b, e1g, e3b, e3g, x_aff)
                ctx.cfg = (d_s, d_1, d_3, training, bypass, True)
                ctx.small_prm = (aff[0], inv, aff[1])
                ctx.mark_non_differentiable(aff)
                _mark_lazy(ctx, raw_e, lazy_ok)
                return raw_e, aff
            out = _new((N, CE, H, W), x)
            gap = _new((N, CE), x) if want_gap else None
            prm = _new((3, CE), x)
            bn_fwd(raw_e, CE, 0, N, CE, E1, H * W, sets[0], sets[1], eps, momentum, prm, out, CE, 0, True,
                   residual=res, r_ctot=Cin, r_coff=0, r_aff=x_aff if bypass else None, gap_out=gap, gap_ctot=CE, gap_coff=0)
            ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, prm[:, :E1], prm[:, E1:],
                                  sb, sg, e1b, e1g, e3b, e3g, x_aff)
            ctx.cfg = (d_s, d_1, d_3, training, bypass, False)
            ctx.small_prm = (prm[0], prm[1], prm[2])
            _mark_lazy(ctx, out, lazy_ok and small)
            if not want_gap:
                return out
            ctx.mark_non_differentiable(gap)
            return out, gap
        ctx.small_prm = None
        if defer:
            aff = _new((3, CE), x)
            inv = _new((CE,), x)
            inv1, inv3 = inv[:E1], inv[E1:]
            ctx.small_prm = (aff[0], inv, aff[1])       # rows over the whole concat buffer: backward may run both layers at once
            d_1, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e1w, e1b, e1g, e1be, e1rm, e1rv, (1, 1), (0, 0), training,
                                  momentum, eps, False, True, raw_e, CE, 0, None, CE, 0, N,
                                  stats_into=(aff[0, :E1], inv1, aff[1, :E1]), shift_into=aff[2, :E1], conv_done=fused)
            d_3, _ = _CBR.forward(act_s, S_, 0, S_, H, W, e3w, e3b, e3g, e3be, e3rm, e3rv, (1, 1), (1, 1), training,
                                  momentum, eps, False, True, raw_e, CE, E1, None, CE, E1, N,
                                  stats_into=(aff[0, E1:], inv3, aff[1, E1:]), shift_into=aff[2, E1:], conv_done=fused)
            ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, aff, inv1, inv3,
                                  sb, sg, e1b, e1g, e3b, e3g, x_aff)
            ctx.cfg = (d_s, d_1, d_3, training, bypass, True)
            ctx.mark_non_differentiable(aff)
            _mark_lazy(ctx, raw_e, lazy_ok)
            return raw_e, aff
        out = _new((N, CE, H, W), x)
        # plane averages of the block output for the SELayer behind it: a by-product of the BN
        # apply kernels in training, one extra pass otherwise
        fused_gap = want_gap and training and _PLANE_BN[0]
        gap = _new((N, CE), x) if fused_gap else None
        d_1, prm_1 = _CBR.forward(act_s, S_, 0, S_, H, W, e1w, e1b, e1g, e1be, e1rm, e1rv, (1, 1), (0, 0), training,
                                  momentum, eps, False, True, raw_e, CE, 0, out, CE, 0, N, res, Cin, 0, gap, CE, 0,
                                  r_aff=x_aff if bypass else None, conv_done=fused)
        d_3, prm_3 = _CBR.forward(act_s, S_, 0, S_, H, W, e3w, e3b, e3g, e3be, e3rm, e3rv, (1, 1),
                                  (1, 1), training, momentum, eps, False, True, raw_e, CE, E1, out,
                                  CE, E1, N, res, Cin, E1, gap, CE, E1, r_aff=x_aff if bypass else None, conv_done=fused)
        ctx.save_for_backward(x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, prm_1,
                              prm_3, sb, sg, e1b, e1g, e3b, e3g, x_aff)
        ctx.cfg = (d_s, d_1, d_3, training, bypass, False)
        if not want_gap:
            return out
        if gap is None:
            gap = ops.gap_fwd(out, N, CE, 0, CE, H * W)
        ctx.mark_non_differentiable(gap)      # the SELayer differentiates through `out` itself
        return out, gap

    @staticmethod
    def backward(ctx, dout, *_unused_dgap):
        d_s, d_1, d_3, training, bypass, deferred = ctx.cfg
        if deferred:
            (x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, aff, inv1, inv3, sb, sg, e1b, e1g,
             e3b, e3g, x_aff) = ctx.saved_tensors
            E1 = e1w.shape[0]
            prm_1 = (aff[0, :E1], inv1, aff[1, :E1])
            prm_3 = (aff[0, E1:], inv3, aff[1, E1:])
        else:
            (x, sw, sbe, e1w, e1be, e3w, e3be, raw_s, act_s, raw_e, prm_s, prm_1, prm_3, sb, sg, e1b, e1g,
             e3b, e3g, x_aff) = ctx.saved_tensors
        N, Cin, H, W = x.shape
        S_, E1, E3 = sw.shape[0], e1w.shape[0], e3w.shape[0]
        CE = E1 + E3
        small_prm = getattr(ctx, "small_prm", None)
        bcoop = ops.bn_coop_ok(N, H * W)
        # a lazy gradient (_LAZY): routed out of the pooled gradient by the cooperative BatchNorm backward; a bypass block hands
        # the entry on with its squeeze data gradient -- when neither is possible here the tensor is completed first
        lazy = _lazy_take(dout, getattr(ctx, "lazy_token", None))
        poolfuse = getattr(ctx, "poolfuse", None)
        if poolfuse is not None and lazy is None:
            raise RuntimeError("a Fire block that pooled its own output (poolfuse) received a gradient that did not come from the "
                               "SELayer + MaxPool node behind it")
        if lazy is not None:
            takes = (small_prm is not None and training and bcoop and ops.bn_coop_pool_ok(N, H, W, lazy.sh)
                     and (not bypass or getattr(ctx, "x_token", None) is not None))
            if takes:
                del _LAZY[dout.data_ptr()]
            else:
                dout = lazy_materialize(dout, lazy)
                lazy = None
        dout = dout.contiguous()
        dact_s = _new((N, S_, H, W), x)
        draw1 = _new((N, E1, H, W), x)
        bg1 = bg3 = amax3 = None
        if small_prm is not None and training and dout.is_cuda and (bcoop or ops.bn_small_ok(N, H * W)):
            # both expand BatchNorms' backward in one launch (dout and the raw concat buffer read once)
            draw3 = _new((N, E3, H, W), x)
            sk = [_sink(p, (E1 if i < 2 else E3,), dout) for i, p in enumerate((e1g, e1be, e3g, e3be))]
            if len({k[1] for k in sk}) > 1:          # one accumulate flag serves the four outputs
                fresh = [_new((E1 if i < 2 else E3,), dout) for i in range(4)]
                sk = [(t, False, t) for t in fresh]
            # (the data / weight gradients on the two-piece kernels take their scale from the largest |draw|)
            amax3 = (ops.amax_slot(dout.device) if (getattr(d_3, "wh2_1", None) is not None
                                                     or getattr(d_1, "wh2_1", None) is not None
                                                     or getattr(d_s, "out_bound", None) is not None) else None)
            if not bcoop and not _SMALL_H2[0]:
                amax3 = None
            if bcoop:
                if lazy is not None:
                    ops.bn_coop_bwd_pool(dout if lazy.stored else None, CE, 0, lazy.pool(), raw_e, CE, 0, small_prm, e1be, e3be, draw1,
                                         draw3, sk[0][0], sk[1][0], sk[2][0], sk[3][0], sk[0][1], N, CE, E1, H, W, True,
                                         amax_out=amax3)
                else:
                    ops.bn_coop_bwd(dout, CE, 0, raw_e, CE, 0, small_prm, e1be, e3be, draw1, draw3, sk[0][0], sk[1][0],
                                    sk[2][0], sk[3][0], sk[0][1], N, CE, E1, H * W, True, amax_out=amax3)
            else:
                ops.bn_small_bwd(dout, CE, 0, raw_e, CE, 0, small_prm, e1be, e3be, draw1, draw3, sk[0][0], sk[1][0], sk[2][0],
                                 sk[3][0], sk[0][1], N, CE, E1, H * W, True, amax_out=amax3)
            bg1, bg3 = (sk[0][2], sk[1][2]), (sk[2][2], sk[3][2])
        wdg = getattr(ctx, "wdg", None)
        g1 = _CBR.backward(dout, CE, 0, act_s, d_1, e1w, e1b, e1g, prm_1, e1be, raw_e, training, False,
                           True, draw1, wdg is None, dact_s, S_, 0, bn_grads=bg1, amax=amax3)
        if bg3 is None:
            draw3 = _new((N, E3, H, W), x)
        g3 = _CBR.backward(dout, CE, E1, act_s, d_3, e3w, e3b, e3g, prm_3, e3be, raw_e, training, False,
                           True, draw3, wdg is None, dact_s, S_, 0, dx_accumulate=True, bn_grads=bg3, amax=amax3,
                           x_bound=getattr(d_s, "out_bound", None))
        if wdg is not None:
            # dS = W3^T * dE3 + W1^T dE1 in one launch: the expand1x1 gradient's channels are centre-tap chunks of the 3x3
            # data-gradient kernel (no second launch, no accumulate pass over dS)
            gd = ops.conv_desc(N, E3, H, W, S_, 3, 3, 1, 1, 1, 1, OH=H, OW=W, in_ctot=E3, in_coff=0, out_ctot=S_, out_coff=0)
            ops.fire_expand_dgrad(draw3, wdg[0], draw1, wdg[1], dact_s, gd)
        del draw1, draw3
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        draw_s = _new((N, S_, H, W), x)
        # the bypass gradient: a lazy one travels on with dx (stored part: dout itself when it holds one)
        res = (dout if (bypass and (lazy is None or lazy.stored)) else None)
        gs = _CBR.backward(dact_s, S_, 0, x, d_s, sw, sb, sg, prm_s, sbe, raw_s, training, False, True,
                           draw_s, need_dx, dx, Cin, 0, res, CE, 0, in_aff=x_aff)
        if lazy is not None and bypass and need_dx:
            # the bypass part of this block's input gradient stays unwritten: the entry travels on with dx (which holds the
            # squeeze data gradient) to the block in front
            _lazy_put(_LazyGrad(dx, lazy.dyp, lazy.idx, lazy.xs, lazy.xadd, lazy.sh, True, ctx.x_token, lazy.full))
        return (dx, gs[0], gs[1], gs[2], gs[3], None, None, g1[0], g1[1], g1[2], g1[3], None, None,
                g3[0], g3[1], g3[2], g3[3], None, None, None, None, None, None, None, None, None, None)


class ConvAddFn(Function):
    """addend + conv1x1(x) + bias: the 'complex' Fire bypass (pointseg_modules.py:110-112,136-138),
    the sum is the residual operand of the convolution's epilogue"""

    @staticmethod
    def forward(ctx, x, weight, bias, addend):
        x, addend = x.contiguous(), addend.contiguous()
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        d = ops.conv_desc(N, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, res_ctot=Cout, res_coff=0)
        y = _new((N, Cout, H, W), x)
        ops.conv2d_fwd(x, ops.conv2d_prepped(weight, 0), bias, y, d, residual=addend)
        d.wt2 = ops.conv2d_prepped(weight, 1)
        ctx.save_for_backward(x, weight, bias)
        ctx.d = d
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        d = ctx.d
        dy = dy.contiguous()
        N, Cout, HW = d.N, d.Cout, d.OH * d.OW
        ret_b = None
        if bias is not None:
            db, acc_b, ret_b = _sink(bias, (Cout,), dy)
            ops.chan_sum(dy, N, Cout, 0, Cout, HW, out=db, accumulate=acc_b)
        dw, acc_w, ret_w = _sink(weight, weight.shape, dy)
        dd = ops.conv_desc(N, d.Cin, d.H, d.W, Cout, 1, 1, 1, 1, 0, 0)
        ops.conv2d_wgrad(x, dy, dw, dd, accumulate=acc_w)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            conv_dgrad(dy, weight, d, dx, d.Cin, 0)
        return dx, ret_w, ret_b, dy


# =============================================================================== pooling / SE
class MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, ceil_mode):
        amax = getattr(x, "_dlio_amax", None)
        x = x.contiguous()
        y, idx = ops.maxpool2d_fwd(x, k, stride[0], stride[1], pad[0], pad[1], ceil_mode)
        ctx.save_for_backward(idx)
        ctx.cfg = (tuple(x.shape), k, stride, pad)
        y._dlio_amax = amax                 # (max |pool(x)| <= max |x|: still a valid operand scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        shape, k, stride, pad = ctx.cfg
        dx = ops.maxpool2d_bwd(dy.contiguous(), idx, shape, k, stride[0], stride[1], pad[0], pad[1])
        return dx, None, None, None, None


class SEPoolFn(Function):
    """SELayer (pointseg_modules.py:216-221) fused with the MaxPool2d that always follows it in
    PSEncoder (pointseg_net.py:27-46): the channel re-weighting is applied while pooling, so the
    scaled full-resolution tensor is never written.  pool=None gives the plain SELayer.
    pooled = (arg-max map, H, W): x is already the POOLED maximum of the block output (FireFn's poolfuse branch; the scale
    s = sigmoid(..) > 0 commutes with the maximum) -- the forward is the fc pair + one pass over the pooled tensor."""

    @staticmethod
    def forward(ctx, x, w1, w2, pool, gap=None, pooled=None):
        tok = getattr(x, "_dlio_lazy_token", None)
        plain = x._backward_hooks is None and not x.retains_grad      # (hooks / retain_grad would see the unwritten tensor)
        x = x.contiguous()
        if pooled is not None:
            idx, H, W = pooled
            N, C_ = x.shape[0], x.shape[1]
            if gap is None or pool is None or tok is None or not plain:
                raise ValueError("a pre-pooled SELayer input needs the plane averages, the pool geometry and an untouched "
                                 "producer output (no hooks / retain_grad)")
        else:
            N, C_, H, W = x.shape
        g = gap if gap is not None else ops.gap_fwd(x, N, C_, 0, C_, H * W)
        if _SE_FC[0] and x.is_cuda and w1.is_contiguous() and w2.is_contiguous() and ops.se_fc_ok(N, C_, w1.shape[0]):
            h, s = ops.se_fc_fwd(g.contiguous(), w1, w2)       # both layers + activations in one launch (se_fc.hip)
        else:
            h = ops.linear_fwd(g, w1, None, ops.ACT_RELU)
            s = ops.linear_fwd(h, w2, None, ops.ACT_SIGMOID)
        if pooled is not None:
            y = ops.chan_scale_fwd(x, s)
        elif pool is None:
            y, idx = ops.chan_scale_fwd(x, s), None
        else:
            k, stride, pad = pool
            y, idx = ops.maxpool2d_fwd(x, k, stride[0], stride[1], pad[0], pad[1], False, x_scale=s)
        # (the pooled output is kept: the scale gradient is sum dy * y / s over POOLED planes, see backward)
        keep_y = pooled is not None or (pool is not None and _SE_POOLED_DOT[0])
        ctx.save_for_backward(None if pooled is not None else x, w1, w2, g, h, s, idx, y if keep_y else None)
        ctx.pool = pool
        ctx.shape = (N, C_, H, W)
        ctx.pre_pooled = pooled is not None
        ctx.lazy_token = tok
        # the producer of x (a Fire block on the cooperative BatchNorm path) routes the pooled gradient itself: _LAZY
        ctx.lazy_ok = (pooled is not None or
                       (pool is not None and tok is not None and plain and _pool_row_stride(pool) != 0
                        and ops.bn_coop_pool_ok(N, H, W, pool[1][0])))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, g, h, s, idx, y = ctx.saved_tensors
        N, C_, H, W = ctx.shape
        dy = dy.contiguous()
        pre = ctx.pre_pooled
        fused = False
        if ctx.pool is None:
            dx, ds = ops.chan_scale_bwd(dy, x, s)
        else:
            k, stride, pad = ctx.pool
            fused = pre or ops.pool_fast_path(H, W, dy.shape[2], dy.shape[3], k, stride[0], stride[1], pad[0], pad[1])
            if fused:
                # two passes over the full-resolution tensor instead of six: ds from (dy, idx, x) with
                # the pooled gradient recomputed on the fly; dx written once at the end
                if y is not None:
                    # y = maxpool(x * s), s = sigmoid > 0: sum_o dy[o] * x[argmax(o)] = sum_o dy[o] * y[o] / s -- two POOLED
                    # tensors instead of the full-resolution x and the arg-max map
                    ds = ops.plane_dot(dy, y, s)
                else:
                    ds = ops.maxpool2d_bwd_dot(dy, idx, x, k, stride[0], stride[1], pad[0], pad[1])
            else:
                dxs = ops.maxpool2d_bwd(dy, idx, tuple(x.shape), k, stride[0], stride[1], pad[0], pad[1])
                dx, ds = ops.chan_scale_bwd(dxs, x, s)
        dw2, acc2, ret2 = _sink(w2, w2.shape, dy)
        dw1, acc1, ret1 = _sink(w1, w1.shape, dy)
        if (_SE_FC[0] and fused and dy.is_cuda and acc1 == acc2 and w1.is_contiguous() and w2.is_contiguous()
                and ops.se_fc_ok(N, C_, w1.shape[0])):
            # sigmoid', W2^T, relu', W1^T and the 1 / (H W) of the plane average's gradient in one launch, both weight
            # gradients in a second one (instead of seven dense launches)
            dgs = ops.se_fc_bwd(ds.contiguous(), s, h, g, w1, w2, dw1, dw2, acc1, 1.0 / (H * W))
        else:
            dz2 = ops.act_bwd(ds, s, ops.ACT_SIGMOID)
            ops.linear_bwd_weight(dz2, h, N, w2.shape[0], w2.shape[1], dw=dw2, want_bias=False, accumulate=acc2)
            dh = ops.linear_bwd_data(dz2, w2, N)
            dz1 = ops.act_bwd(dh, h, ops.ACT_RELU)
            ops.linear_bwd_weight(dz1, g, N, w1.shape[0], w1.shape[1], dw=dw1, want_bias=False, accumulate=acc1)
            dg = ops.linear_bwd_data(dz1, w1, N)
            if not fused:
                ops.gap_bwd(dg, dx, N, C_, H * W, accumulate=True)
                return dx, ret1, ret2, None, None, None
            dgs = ops.ew_scale(dg, 1.0 / (H * W))
        k, stride, pad = ctx.pool
        if pre or (getattr(ctx, "lazy_ok", False) and _LAZY_POOL[0] and ops._SYNC_BN[0] is None):
            # NOT written here: the Fire block's BatchNorm backward routes dy itself.  pre-pooled input: the tensor handed
            # back has the pooled shape (it stands for the gradient of the block output, which was never stored either)
            dx = torch.empty(dy.shape if pre else (N, C_, H, W), dtype=torch.float32, device=dy.device)
            _lazy_put(_LazyGrad(dx, dy, idx, s, dgs, stride[0], False, ctx.lazy_token, (N, C_, H, W)))
            return dx, ret1, ret2, None, None, None
        dx = ops.maxpool2d_bwd(dy, idx, (N, C_, H, W), k, stride[0], stride[1], pad[0], pad[1], x_scale=s, x_add=dgs)
        return dx, ret1, ret2, None, None, None


class GapFn(Function):
    """adaptive_avg_pool2d(x, (1,1)).flatten(1)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        N, C_, H, W = x.shape
        ctx.shape = (N, C_, H, W)
        return ops.gap_fwd(x, N, C_, 0, C_, H * W)

    @staticmethod
    def backward(ctx, dy):
        N, C_, H, W = ctx.shape
        dx = _new(ctx.shape, dy)
        ops.gap_bwd(dy.contiguous(), dx, N, C_, H * W)
        return dx


# =============================================================================== small dense ops
class LinearFn(Function):
    """act(x W^T + b) over the last dimension."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x.contiguous()
        K = w.shape[1]
        M = x.numel() // K
        y = ops.linear_fwd(x, w, b, act, M=M)
        ctx.save_for_backward(x, w, y, b)
        ctx.cfg = (act, M)
        return y.view(x.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x, w, y, b = ctx.saved_tensors
        act, M = ctx.cfg
        dy = dy.contiguous()
        dz = ops.act_bwd(dy, y, act) if act else dy
        dx = ops.linear_bwd_data(dz, w, M).view(x.shape) if ctx.needs_input_grad[0] else None
        dw, acc_w, ret_w = _sink(w, w.shape, dy)
        db = ret_b = None
        if b is not None:
            db, acc_b, ret_b = _sink(b, b.shape, dy)
            if acc_b != acc_w:       # one accumulate flag per launch: fall back to fresh buffers
                dw, acc_w, ret_w = _new(tuple(w.shape), dy), False, None
                db, ret_b = _new(tuple(b.shape), dy), None
                ret_w, ret_b = dw, db
        ops.linear_bwd_weight(dz, x, M, w.shape[0], w.shape[1], dw=dw, db=db, want_bias=b is not None,
                              accumulate=acc_w)
        return dx, ret_w, ret_b, None


class PairFuseFcFn(Function):
    """pair_fuse_fc: act(fc1(gap(a) (+|-) gap(b))) of the two encoder outputs (lidar_feat_nets.py:84-94, :131-141) as one
    launch (csrc/pair_fuse.hip); backward: activation, weight gradient, data gradient, one broadcast launch for both maps"""

    @staticmethod
    def forward(ctx, a, b, mode, w, bias, act):
        a, b = a.contiguous(), b.contiguous()
        feat, y = ops.pair_fuse_fc_fwd(a, b, mode, w, bias, act)
        ctx.save_for_backward(feat, w, y, bias)
        ctx.cfg = (mode, act, tuple(a.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        feat, w, y, bias = ctx.saved_tensors
        mode, act, shape = ctx.cfg
        dy = dy.contiguous()
        M = feat.shape[0]
        dz = ops.act_bwd(dy, y, act) if act else dy
        dw, acc_w, ret_w = _sink(w, w.shape, dy)
        db = ret_b = None
        if bias is not None:
            db, acc_b, ret_b = _sink(bias, bias.shape, dy)
            if acc_b != acc_w:
                dw, acc_w, ret_w = _new(tuple(w.shape), dy), False, None
                db, ret_b = _new(tuple(bias.shape), dy), None
                ret_w, ret_b = dw, db
        ops.linear_bwd_weight(dz, feat, M, w.shape[0], w.shape[1], dw=dw, db=db, want_bias=bias is not None, accumulate=acc_w)
        da = dbm = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            da, dbm = ops.pair_fuse_bwd(ops.linear_bwd_data(dz, w, M), shape, mode)
        return (da if ctx.needs_input_grad[0] else None, dbm if ctx.needs_input_grad[1] else None, None, ret_w, ret_b, None)


class BinaryFn(Function):
    """op: 0 a+b, 1 a-b, 2 a*b, 3 relu(a+b)"""

    @staticmethod
    def forward(ctx, a, b, op):
        a, b = a.contiguous(), b.contiguous()
        # relu(a + b), the BasicBlock tail: its consumer is a 3x3 convolution -- leave the largest magnitude for it
        slot = (ops.amax_slot_kept(a.device) if (op == 3 and a.is_cuda and _CONV_H2_FWD[0] and a.numel() % 4 == 0
                                                   and any(ctx.needs_input_grad[:2]) and ops._SYNC_BN[0] is None) else None)
        y = ops.ew_binary(a, b, op, amax_out=slot)
        y._dlio_amax = slot
        ctx.op = op
        if op == 2:
            ctx.save_for_backward(a, b)
        elif op == 3:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        op = ctx.op
        if op == 0:
            return dy, dy, None
        if op == 1:
            return dy, ops.ew_scale(dy, -1.0), None
        if op == 2:
            a, b = ctx.saved_tensors
            return ops.ew_binary(dy, b, 2), ops.ew_binary(dy, a, 2), None
        (y,) = ctx.saved_tensors
        g = ops.act_bwd(dy, y, ops.ACT_RELU)
        return g, g, None


class Cat2Fn(Function):
    """torch.cat((a, b), dim=-1) for [.., Fa] / [.., Fb]"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        Fa, Fb = a.shape[-1], b.shape[-1]
        rows = a.numel() // Fa
        out = _new(a.shape[:-1] + (Fa + Fb,), a)
        ops.copy2d(a, Fa, out, Fa + Fb, rows, Fa)
        ops.copy2d(b, Fb, out, Fa + Fb, rows, Fb, dst_off=Fa)
        ctx.cfg = (tuple(a.shape), tuple(b.shape), rows)
        return out

    @staticmethod
    def backward(ctx, dy):
        sa, sb, rows = ctx.cfg
        dy = dy.contiguous()
        Fa, Fb = sa[-1], sb[-1]
        da, db = _new(sa, dy), _new(sb, dy)
        ops.copy2d(dy, Fa + Fb, da, Fa, rows, Fa)
        ops.copy2d(dy, Fa + Fb, db, Fb, rows, Fb, src_off=Fa)
        return da, db


class SegSumFn(Function):
    """x [G, R, C] -> sum over R (imu_feat_nets.py:49)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        G, R, C_ = x.shape
        ctx.shape = (G, R, C_)
        return ops.seg_sum_fwd(x, G, R, C_)

    @staticmethod
    def backward(ctx, dy):
        return ops.seg_sum_bwd(dy.contiguous(), *ctx.shape)


_DROPOUT_STATE = {"seed": 0x5EED, "offset": 0}


def manual_seed(seed):
    """Seed of the Philox stream used by every dropout on the path (one stream per process;
    the counter advances by the number of generated groups)."""
    _DROPOUT_STATE["seed"] = int(seed)
    _DROPOUT_STATE["offset"] = 0


def dropout_offset():
    return _DROPOUT_STATE["offset"]


def _dropout_launch(x, p):
    off = _DROPOUT_STATE["offset"]
    _DROPOUT_STATE["offset"] = off + (x.numel() + 3) // 4
    return ops.dropout_fwd(x, p, _DROPOUT_STATE["seed"], off)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        y, mask = _dropout_launch(x, p)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.dropout_bwd(dy.contiguous(), mask, ctx.p), None


def dropout(x, p, training):
    if not training or p <= 0.:
        return x
    return DropoutFn.apply(x, p)


# =============================================================================== recurrent nets
class RNNFn(Function):
    """Multi-layer (bi)directional LSTM/GRU over `Sg` consecutive sub-sequences whose state is
    carried from one to the next (ImufeatRNN0.forward, imu_feat_nets.py:75-83; with Sg=1 it is
    a plain nn.LSTM/nn.GRU call as in OdomFeatRNN.forward, odom_feat_nets.py:72-83).

    x [B, Sg, T, I] -> top-layer outputs [B, Sg, T, D*H].  Weights per (layer, direction):
    w_ih, w_hh, b_ih, b_hh in nn.LSTM order.  Inter-layer dropout as in nn.LSTM (train only)."""

    @staticmethod
    def forward(ctx, x, mode, H, L, D, p, training, *weights):
        x = x.contiguous()
        B, Sg, T, I = x.shape
        G = 4 if mode == "lstm" else 3
        dev = x
        rows = B * T
        W = [[weights[(l * D + d) * 4:(l * D + d) * 4 + 4] for d in range(D)] for l in range(L)]
        state_h = [[None] * D for _ in range(L)]
        state_c = [[None] * D for _ in range(L)]
        saved = []          # per segment: per layer: dict
        tops = _new((B, Sg, T, D * H), x)
        side = _rnn_dir_stream(x) if D == 2 else None
        cur = current_stream_obj() if side is not None else None
        for s in range(Sg):
            inp = x[:, s].contiguous().view(rows, I)
            seg = []
            for l in range(L):
                out_l = _new((rows, D * H), dev)
                rec = {"inp": inp, "dirs": []}
                # every buffer is allocated on the caller's stream; only launches move to the companion
                bufs = []
                for d in range(D):
                    bb = {"gx": _new((rows, G * H), dev), "hp": _new((rows, H), dev), "gates": _new((rows, 4 * H), dev),
                          "hT": _new((B, H), dev)}
                    if mode == "lstm":
                        bb["cs"], bb["cT"] = _new((rows, H), dev), _new((B, H), dev)
                    bufs.append(bb)
                if side is not None:
                    side.wait_stream(cur)
                for d in range(D):
                    w_ih, w_hh, b_ih, b_hh = W[l][d]
                    bb = bufs[d]
                    h0 = state_h[l][d]
                    with (on_stream(side) if (side is not None and d == 1) else contextlib.nullcontext()):
                        ops.linear_fwd(inp, w_ih, b_ih, M=rows, out=bb["gx"])
                        if mode == "lstm":
                            c0 = state_c[l][d]
                            # (nobody reads the state behind the last sub-sequence: no copy-out launch there)
                            last = s == Sg - 1
                            ops.lstm_seq_fwd(bb["gx"], w_hh, b_hh, h0, c0, out_l, d * H, D * H, bb["cs"], bb["hp"],
                                             bb["gates"], None if last else bb["hT"], None if last else bb["cT"], T, B, H,
                                             1, T, d == 1)
                            rec["dirs"].append({"gates": bb["gates"], "cs": bb["cs"], "hp": bb["hp"], "c0": c0})
                            state_c[l][d] = bb["cT"]
                        else:
                            ops.gru_seq_fwd(bb["gx"], w_hh, b_hh, h0, out_l, d * H, D * H, bb["hp"], bb["gates"],
                                            bb["hT"], T, B, H, 1, T, d == 1)
                            rec["dirs"].append({"gates": bb["gates"], "hp": bb["hp"]})
                    state_h[l][d] = bb["hT"]
                if side is not None:
                    cur.wait_stream(side)
                    for t in list(bufs[1].values()) + [inp, out_l]:
                        t.record_stream(side)
                rec["out"] = out_l
                if l + 1 < L and training and p > 0.:
                    inp, mask = _dropout_launch(out_l, p)
                    rec["mask"] = mask
                else:
                    inp = out_l
                seg.append(rec)
            ops.copy2d(seg[-1]["out"].view(B, T * D * H), T * D * H, tops, Sg * T * D * H, B, T * D * H,
                       dst_off=s * T * D * H)
            saved.append(seg)
        ctx.saved = saved
        ctx.weights = weights
        ctx.cfg = (mode, H, L, D, p, B, Sg, T, I)
        return tops

    @staticmethod
    def backward(ctx, dtops):
        mode, H, L, D, p, B, Sg, T, I = ctx.cfg
        weights, saved = ctx.weights, ctx.saved
        G = 4 if mode == "lstm" else 3
        rows = B * T
        dtops = dtops.contiguous()
        dev = dtops
        W = [[weights[(l * D + d) * 4:(l * D + d) * 4 + 4] for d in range(D)] for l in range(L)]
        grads = [None] * len(weights)      # what autograd receives (None when sunk into .grad)
        outs = [None] * len(weights)       # where the kernels write
        first = [True] * len(weights)

        def ensure_w(slot, N_, K, with_bias_slot):
            """gradient buffers of one weight / bias pair (allocated on the caller's stream)"""
            wi = slot
            if outs[wi] is None:
                ow, aw_, rw = _sink(weights[wi], (N_, K), dev)
                ob, ab_, rb = _sink(weights[with_bias_slot], (N_,), dev)
                if aw_ != ab_:
                    ow, aw_, rw = _new((N_, K), dev), False, None
                    ob, rb = _new((N_,), dev), None
                    rw, rb = ow, ob
                outs[wi], outs[with_bias_slot] = ow, ob
                grads[wi], grads[with_bias_slot] = rw, rb
                # sunk gradients accumulate from the first call on -- unless their slots are of the kind nobody zeroes
                # (FlatOptimizer.set_overwritten): then this pass's first write replaces what is there
                first[wi] = (not aw_) or bool(getattr(weights[wi], "_dlio_grad_overwrite", False))

        def acc_w(slot, dz, lddz, xin, ldx, N_, K, with_bias_slot):
            wi = slot
            # (forking these onto the weight-gradient stream was measured: 34.7 vs 33.5 ms/step)
            ops.linear_bwd_weight(dz, xin, rows, N_, K, dw=outs[wi], db=outs[with_bias_slot],
                                  lddz=lddz, ldx=ldx, accumulate=not first[wi])
            first[wi] = False

        dstate_h = [[None] * D for _ in range(L)]
        dstate_c = [[None] * D for _ in range(L)]
        dx = _new((B, Sg, T, I), dev) if ctx.needs_input_grad[0] else None
        side = _rnn_dir_stream(dtops) if D == 2 else None
        cur = current_stream_obj() if side is not None else None
        for s in reversed(range(Sg)):
            seg = saved[s]
            dout = _new((rows, D * H), dev)
            ops.copy2d(dtops, Sg * T * D * H, dout.view(B, T * D * H), T * D * H, B, T * D * H,
                       src_off=s * T * D * H)
            for l in reversed(range(L)):
                rec = seg[l]
                K_in = rec["inp"].shape[1]
                need_dinp = l > 0 or dx is not None
                dinp = _new((rows, K_in), dev) if need_dinp else None
                bufs = []
                for d in range(D):
                    # gradient w.r.t. the state a segment started from: wanted by the segment before
                    # it; segment 0 started from zeros
                    bb = {"dh0": _new((B, H), dev) if s > 0 else None}
                    if mode == "lstm":
                        bb["dg"] = _new((rows, 4 * H), dev)
                        bb["dc0"] = _new((B, H), dev) if s > 0 else None
                    else:
                        bb["dgx"], bb["dgh"] = _new((rows, 3 * H), dev), _new((rows, 3 * H), dev)
                    bufs.append(bb)
                    ensure_w((l * D + d) * 4 + 0, G * H, K_in, (l * D + d) * 4 + 2)
                    ensure_w((l * D + d) * 4 + 1, G * H, H, (l * D + d) * 4 + 3)
                if side is not None:
                    side.wait_stream(cur)
                for d in range(D):
                    w_ih, w_hh, b_ih, b_hh = W[l][d]
                    base = (l * D + d) * 4
                    sv = rec["dirs"][d]
                    bb = bufs[d]
                    # the two directions are independent until their input gradients are summed: the
                    # reverse direction's recurrence and weight gradients run on a companion stream
                    with (on_stream(side) if (side is not None and d == 1) else contextlib.nullcontext()):
                        if mode == "lstm":
                            ops.lstm_seq_bwd(dout, d * H, D * H, dstate_h[l][d], dstate_c[l][d],
                                             sv["gates"], sv["cs"], sv["c0"], w_hh, bb["dg"], bb["dh0"], bb["dc0"], T, B,
                                             H, 1, T, d == 1)
                            bb["dgx"] = bb["dgh"] = bb["dg"]
                            dstate_c[l][d] = bb["dc0"]
                        else:
                            ops.gru_seq_bwd(dout, d * H, D * H, dstate_h[l][d], sv["gates"], sv["hp"],
                                            w_hh, bb["dgx"], bb["dgh"], bb["dh0"], T, B, H, 1, T, d == 1)
                        dstate_h[l][d] = bb["dh0"]
                        acc_w(base + 0, bb["dgx"], G * H, rec["inp"], K_in, G * H, K_in, base + 2)
                        acc_w(base + 1, bb["dgh"], G * H, sv["hp"], H, G * H, H, base + 3)
                if side is not None:
                    cur.wait_stream(side)
                    for t in [v for v in bufs[1].values() if v is not None] + [dout]:
                        t.record_stream(side)
                if need_dinp:
                    for d in range(D):
                        ops.linear_bwd_data(bufs[d]["dgx"], W[l][d][0], rows, out=dinp, accumulate=d > 0)
                if l > 0:
                    prev = seg[l - 1]
                    dout = ops.dropout_bwd(dinp, prev["mask"], p) if "mask" in prev else dinp
                elif dx is not None:
                    ops.copy2d(dinp.view(B, T * I), T * I, dx, Sg * T * I, B, T * I, dst_off=s * T * I)
        ctx.saved = None
        return (dx, None, None, None, None, None, None) + tuple(grads)


_TAIL_FUSED = [os.environ.get("DLIO_TAIL_FUSED", "1") != "0"]      # soft fusion / dropout + heads as one launch each


def _rows_view(t):
    """t [..., F] -> (tensor to hand to a kernel, row stride) when its rows are F contiguous floats a uniform stride apart
    (a contiguous tensor, or a slice like rnn_out[:, :, -1, :H]); else a contiguous copy"""
    F_ = t.shape[-1]
    if t.stride(-1) == 1 and t.dim() >= 2:
        ld, ok = t.stride(-2), True
        for i in range(t.dim() - 2):            # leading dimensions must collapse onto the row stride
            ok = ok and t.stride(i) == t.stride(i + 1) * t.shape[i + 1]
        if ok and ld >= F_ and ld % 4 == 0 and t.data_ptr() % 16 == 0:
            return t, ld
    t = t.contiguous()
    return t, F_


class SoftFusionFn(Function):
    """DeepLIOFusionSoft.forward (fusion_nets.py:64-75) as one launch each way: -> (out [.., Fa + Fb], gate [.., Fa + Fb] = [s1 | s2])"""

    @staticmethod
    def forward(ctx, a, b, w1, b1, w2, b2):
        lead = tuple(a.shape[:-1])
        Fa, Fb = a.shape[-1], b.shape[-1]
        R = a.numel() // Fa
        av, lda = _rows_view(a)
        bv, ldb = _rows_view(b)
        out, gate = ops.soft_fusion_fwd(av, lda, bv, ldb, R, Fa, Fb, w1, b1, w2, b2)
        ctx.save_for_backward(av, bv, gate, w1, w2, b1, b2)
        ctx.cfg = (lead, R, Fa, Fb, lda, ldb)
        ctx.mark_non_differentiable(gate)
        return out.view(lead + (Fa + Fb,)), gate.view(lead + (Fa + Fb,))

    @staticmethod
    def backward(ctx, dout, _dgate):
        av, bv, gate, w1, w2, b1, b2 = ctx.saved_tensors
        lead, R, Fa, Fb, lda, ldb = ctx.cfg
        dout = dout.contiguous().view(gate.shape)
        sinks = [_sink(w1, w1.shape, dout), _sink(b1, b1.shape, dout), _sink(w2, w2.shape, dout), _sink(b2, b2.shape, dout)]
        acc = all(sk[1] for sk in sinks)
        if not acc and any(sk[1] for sk in sinks):
            fresh = [_new(tuple(t.shape), dout) for t in (w1, b1, w2, b2)]
            sinks = [(t, False, t) for t in fresh]
        da, db = ops.soft_fusion_bwd(dout, av, lda, bv, ldb, R, Fa, Fb, gate, w1, w2, sinks[0][0], sinks[1][0], sinks[2][0],
                                     sinks[3][0], acc)
        return (da.view(lead + (Fa,)), db.view(lead + (Fb,)), sinks[0][2], sinks[1][2], sinks[2][2], sinks[3][2])


class HeadsFn(Function):
    """DeepLIO.forward's end (deeplio_nets.py:84-90): dropout(p) -> fc_pos, fc_ori, as one launch each way.  x [B, S, ldx]: the
    first K = fc_pos.in_features columns of every row are the features (the forward half of the odometry LSTM's bidirectional
    output, read in place); its gradient comes back full width, the other columns zero."""

    @staticmethod
    def forward(ctx, x, wp, bp, wo, bo, p, training):
        x = x.contiguous()
        K = wp.shape[1]
        ldx = x.shape[-1]
        R = x.numel() // ldx
        drop = training and p > 0.
        off = _DROPOUT_STATE["offset"]
        if drop:
            _DROPOUT_STATE["offset"] = off + (R * K + 3) // 4
        pos, ori, mask = ops.heads_fwd(x, ldx, R, K, wp, bp, wo, bo, p if drop else 0., _DROPOUT_STATE["seed"], off)
        ctx.save_for_backward(x, wp, wo, bp, bo, mask)
        ctx.cfg = (R, K, ldx, p if drop else 0.)
        lead = x.shape[:-1]
        return pos.view(lead + (3,)), ori.view(lead + (3,))

    @staticmethod
    def backward(ctx, dpos, dori):
        x, wp, wo, bp, bo, mask = ctx.saved_tensors
        R, K, ldx, p = ctx.cfg
        dpos = (dpos if dpos is not None else torch.zeros(R, 3, device=x.device)).contiguous()
        dori = (dori if dori is not None else torch.zeros(R, 3, device=x.device)).contiguous()
        sinks = [_sink(wp, wp.shape, x), _sink(bp, bp.shape, x), _sink(wo, wo.shape, x), _sink(bo, bo.shape, x)]
        acc = all(sk[1] for sk in sinks)
        if not acc and any(sk[1] for sk in sinks):
            fresh = [_new(tuple(t.shape), x) for t in (wp, bp, wo, bo)]
            sinks = [(t, False, t) for t in fresh]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ops.heads_bwd(dpos, dori, x, ldx, mask, wp, wo, dx, ldx, sinks[0][0], sinks[1][0], sinks[2][0], sinks[3][0], R, K, p, acc)
        return dx, sinks[0][2], sinks[1][2], sinks[2][2], sinks[3][2], None, None


# the LSTM layer's weight-gradient launch forked onto the companion stream: the serial middle gets 60 us shorter and the step
# 0.1 ms LONGER (median 18.07 against 17.96 ms, eight alternations: the launch then shares the hardware queue with the early
# optimizer sweep and the first convolution weight gradients) -- off
_LSTM_WGRAD_FORK = [os.environ.get("DLIO_LSTM_WGRAD_FORK", "0") != "0"]
_LSTM_LAYER = [os.environ.get("DLIO_LSTM_LAYER", "1") != "0"]      # wide LSTMs: one launch sequence per LAYER (lstm_stream.hip)


def lstm_stack_ok(x4, mode, H, L, D):
    """RNNParams.run may take LstmStackFn: an LSTM over ONE sequence per sample (no state carried between sub-sequences), wide
    enough for the weight-streaming layer kernels"""
    if not (_LSTM_LAYER[0] and mode == "lstm" and x4.is_cuda and x4.dim() == 4 and x4.shape[1] == 1):
        return False
    B, _, T, I = x4.shape
    return all(ops.lstm_layer_ok(T, B, I if l == 0 else D * H, H, D) for l in range(L))


class LstmStackFn(Function):
    """nn.LSTM(I -> H, L layers, D directions, batch_first) over one sequence per sample from the zero state -- the odometry
    net (odom_feat_nets.py:61-68,72-83: 256 -> 1024, 2 layers, bidirectional, over the S axis): every layer is ONE call that
    runs both directions and all steps (csrc/lstm_stream.hip: 2 T launches forward, 2 T + 2 backward, weights streamed in
    K slices / N slabs at HBM rate) instead of RNNFn's per-direction, per-step dlio_linear_* + cell launches on two streams.
    x [B, T, I] -> [B, T, D H]; inter-layer dropout as nn.LSTM (train only).
    top_fwd_only: the caller keeps the forward direction of the top layer only (odom_feat_nets.py:82, SURVEY Q3: the reverse
    half of a bidirectional top layer is discarded, its parameters get a zero gradient) -- that direction of that layer is not
    run at all (a quarter of the net's weight traffic each way); the result is [B, T, H] and the reverse direction's
    parameters receive no gradient (None: an all-zero gradient for the optimizer, as in the reference)."""

    @staticmethod
    def forward(ctx, x, H, L, D, p, training, top_fwd_only, *weights):
        x = x.contiguous()
        B, T, I = x.shape
        rows = B * T
        W = [[weights[(l * D + d) * 4:(l * D + d) * 4 + 4] for d in range(D)] for l in range(L)]
        inp = x.view(rows, I)
        saved = []
        out = None
        for l in range(L):
            Il = inp.shape[1]
            Dl = 1 if (top_fwd_only and l == L - 1) else D
            out = _new((rows, Dl * H), x)
            cs, hp, gates = ops.lstm_layer_fwd(inp, Il, W[l], out, Dl * H, T, B, Il, H, Dl)
            rec = {"inp": inp, "cs": cs, "hp": hp, "gates": gates, "D": Dl}
            if l + 1 < L and training and p > 0.:
                inp, rec["mask"] = _dropout_launch(out, p)
            else:
                inp = out
            saved.append(rec)
        ctx.saved, ctx.weights = saved, weights
        ctx.cfg = (H, L, D, p, B, T, I)
        return out.view(B, T, out.shape[1])

    @staticmethod
    def backward(ctx, dtop):
        H, L, D, p, B, T, I = ctx.cfg
        weights, saved = ctx.weights, ctx.saved
        rows = B * T
        dout = dtop.contiguous().view(rows, -1)
        grads = [None] * len(weights)
        dx = None
        for l in reversed(range(L)):
            rec = saved[l]
            Dl = rec["D"]
            Il = rec["inp"].shape[1]
            need_dinp = l > 0 or ctx.needs_input_grad[0]
            dinp = _new((rows, Il), dout) if need_dinp else None
            shapes = ((4 * H, Il), (4 * H, H), (4 * H,), (4 * H,))
            sinks = [[_sink(weights[(l * D + d) * 4 + j], shapes[j], dout) for j in range(4)] for d in range(Dl)]
            acc = all(sk[1] for dd in sinks for sk in dd)
            if not acc and any(sk[1] for dd in sinks for sk in dd):     # one accumulate flag per launch: fresh buffers for all
                sinks = [[_new(shapes[j], dout) for j in range(4)] for d in range(Dl)]
                sinks = [[(t, False, t) for t in dd] for dd in sinks]
            elif acc and all(getattr(weights[(l * D + d) * 4 + j], "_dlio_grad_overwrite", False)
                             for d in range(Dl) for j in range(4)):
                # FlatOptimizer.set_overwritten: these slots are not zeroed between steps, this launch is their one writer
                acc = False
            for d in range(Dl):
                for j in range(4):
                    grads[(l * D + d) * 4 + j] = sinks[d][j][2]
            for d in range(Dl, D):                # the direction that was not run: an all-zero gradient (a sunk .grad stays as it is)
                for j in range(4):
                    t, a_, r_ = _sink(weights[(l * D + d) * 4 + j], shapes[j], dout)
                    grads[(l * D + d) * 4 + j] = None if a_ else t.zero_()
            wpairs = [(weights[(l * D + d) * 4], weights[(l * D + d) * 4 + 1]) for d in range(Dl)]
            outs = [tuple(sk[0] for sk in sinks[d]) for d in range(Dl)]
            ws = _wgrad_stream(dout) if (acc and _LSTM_WGRAD_FORK[0]) else None
            if ws is None:
                ops.lstm_layer_bwd(dout, Dl * H, rec["inp"], Il, rec["hp"], rec["gates"], rec["cs"], wpairs, outs, acc, dinp, Il,
                                   T, B, Il, H, Dl)
            else:
                # the weight-gradient launch (33 / 29 us on the step's serial chain) goes straight into the flat gradient buffer and
                # nothing on the tape waits for it: forked onto the companion stream like the convolutions' weight gradients
                dg = ops.lstm_layer_bwd(dout, Dl * H, rec["inp"], Il, rec["hp"], rec["gates"], rec["cs"], wpairs, None, acc, dinp,
                                        Il, T, B, Il, H, Dl)
                inp_l, hp_l = rec["inp"], rec["hp"]
                _forked(ws, lambda: ops.lstm_layer_wgrad(dg, inp_l, Il, hp_l, outs, True, T, B, Il, H, Dl), dg, inp_l, hp_l)
            if l > 0:
                prev = saved[l - 1]
                dout = ops.dropout_bwd(dinp, prev["mask"], p) if "mask" in prev else dinp
            else:
                dx = dinp.view(B, T, I) if dinp is not None else None
        ctx.saved = None
        return (dx, None, None, None, None, None, None) + tuple(grads)


# =============================================================================== pose chain / loss
class SE3ChainFn(Function):
    """Trainer.se3_to_SE3 (trainer.py:324-351): f2f increments -> f2g (p, q).  order 0 = wxyz
    (trainer), 1 = xyzw (tester.py:249).  `status` (int32[1], optional) collects the
    determinant / orthonormality checks without a host sync."""

    @staticmethod
    def forward(ctx, t, w, order, status):
        t, w = t.contiguous(), w.contiguous()
        p, q, R = ops.se3_chain_fwd(t, w, order, status)
        ctx.save_for_backward(t, w, R)
        ctx.order = order
        return p, q

    @staticmethod
    def backward(ctx, dp, dq):
        t, w, R = ctx.saved_tensors
        dt, dw = ops.se3_chain_bwd(t, w, R, dp.contiguous(), dq.contiguous(), ctx.order)
        return dt, dw, None, None


class PoseTailFn(Function):
    """Trainer's tail as ONE tape node: se3_to_SE3 (trainer.py:324-351) + the criterion on (f2f_t, f2f_w, f2g_p[:, g0:g1],
    f2g_q[:, g0:g1]) against the columns of gt_f2f / gt_f2g (trainer.py:245-252) + the non-finite check of the model output
    (trainer.py:240-243): two launches forward, two backward (SE3ChainFn + slices + PoseLossFn + the engine's accumulations
    are 21, all on the serial chain of the step's middle).  Same kernels, same values."""

    @staticmethod
    def forward(ctx, sx, sq, beta, mode, terms, t, w, gt_f2f, gt_f2g, g0, g1, order, status, nonfinite):
        t, w, gt_f2f, gt_f2g = t.contiguous(), w.contiguous(), gt_f2f.contiguous(), gt_f2g.contiguous()
        out, p, q, R = ops.pose_tail_fwd(t, w, gt_f2f, gt_f2g, g0, g1, terms, sx, sq, beta, mode, order, status, nonfinite)
        ctx.saved = (t, w, gt_f2f, gt_f2g, sx, sq, p, q, R, out)
        ctx.cfg = (beta, mode, terms, g0, g1, order)
        ctx.terms_out = out
        return out[0]

    @staticmethod
    def backward(ctx, g):
        t, w, gt_f2f, gt_f2g, sx, sq, p, q, R, out = ctx.saved
        beta, mode, terms, g0, g1, order = ctx.cfg
        ctx.saved = None
        dsx = dsq = ret_x = ret_q = None
        acc = False
        if (mode & 1) == 0 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            dsx, acc_x, ret_x = _sink(sx, (), t)
            dsq, acc_q, ret_q = _sink(sq, (), t)
            if acc_x != acc_q:
                dsx, dsq = _new((), t), _new((), t)
                acc_x, ret_x, ret_q = False, dsx, dsq
            acc = acc_x
        dt, dw = ops.pose_tail_bwd(t, w, gt_f2f, gt_f2g, g0, g1, terms, sx, sq, beta, mode, order, p, q, R, out, g.contiguous(),
                                   dsx, dsq, acc)
        return (ret_x, ret_q, None, None, None, dt, dw) + (None,) * 7


class PoseLossFn(Function):
    """HWSLoss / LWSLoss forward+backward in one launch each (losses/losses.py:21-39,68-86)."""

    @staticmethod
    def forward(ctx, sx, sq, beta, mode, use_local, use_global, pt, pw, pp, pq, gt, gw, gp, gq):
        preds = [pt.contiguous() if use_local else None, pw.contiguous() if use_local else None,
                 pp.contiguous() if use_global else None, pq.contiguous() if use_global else None]
        gts = [gt.contiguous() if use_local else None, gw.contiguous() if use_local else None,
               gp.contiguous() if use_global else None, gq.contiguous() if use_global else None]
        out = ops.pose_loss_fwd(preds, gts, sx, sq, beta, mode)
        ctx.saved = (preds, gts, sx, sq, out)
        ctx.cfg = (beta, mode)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        preds, gts, sx, sq, out = ctx.saved
        beta, mode = ctx.cfg
        dps, dsx, dsq = ops.pose_loss_bwd(preds, gts, sx, sq, beta, mode, out, g.contiguous())
        ctx.saved = None
        return (dsx, dsq, None, None, None, None) + tuple(dps) + (None,) * 4
