"""-m gpu parity tests of the composed layers (Fire, conv+BN(+ReLU), SE(+pool), BasicBlock,
RNN stacks, soft fusion) against the oracle's modules: forward, input gradient and every
parameter gradient at 1e-4 of the tensor scale."""
import os
import sys
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_common as gc  # noqa: E402
from conftest import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ctx(dev, seq=2):
    from deeplio_amd import misc
    from deeplio_amd.config import make_config
    cfg = make_config(seq=seq)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=2))
    return cfg


class _DecisionMargin:
    """Records, during an fp64 oracle forward, how close any ReLU input / max-pool runner-up
    comes to its decision boundary (relative to the tensor's scale).  A gradient comparison is
    only meaningful where no decision can flip under fp32 rounding: ONE flipped ReLU among 1e6
    activations moves a whole weight-gradient row by 1e-2 (measured: tools/relu_flip_probe.py)."""

    def __enter__(self):
        self.margin = float("inf")
        self._relu, self._pool = F.relu, F.max_pool2d

        def relu(x, inplace=False):
            if x.numel():
                nz = x.detach().abs()
                nz = nz[nz > 0]
                if nz.numel():
                    self.margin = min(self.margin, float(nz.min()) / max(float(x.detach().abs().max()), 1e-30))
            return self._relu(x)

        def pool(x, k, stride=None, padding=0, **kw):
            y = self._pool(x, k, stride, padding, **kw)
            pad = padding if isinstance(padding, tuple) else (padding, padding)
            st = stride if isinstance(stride, tuple) else (stride, stride)
            u = F.unfold(F.pad(x.detach(), (pad[1], pad[1], pad[0], pad[0]), value=float("-inf")), k, stride=st)
            u = u.view(x.shape[0], x.shape[1], k * k, -1)
            top2 = u.topk(2, dim=2).values
            gap = (top2[:, :, 0] - top2[:, :, 1])
            gap = gap[(gap > 0) & torch.isfinite(gap)]
            if gap.numel():
                self.margin = min(self.margin, float(gap.min()) / max(float(x.detach().abs().max()), 1e-30))
            return y

        F.relu, F.max_pool2d = relu, pool
        return self

    def __exit__(self, *a):
        F.relu, F.max_pool2d = self._relu, self._pool


def compare(hip, ora, x, dev, train=True, tol=TOL, fwd=None, ofwd=None, margin=2e-6):
    """same weights (by key), same input; checks y, dx and all parameter grads against the
    fp64 oracle at `tol` of each tensor's scale.  Weight seeds are advanced until the oracle
    forward is decision-stable (no ReLU input / pool runner-up within `margin` of its boundary),
    so the comparison does not depend on which side of zero fp32 rounding lands."""
    ora.double()
    xo = x.clone().double().requires_grad_(True)
    for seed in range(77, 177):
        gc.fill_state(ora, seed)
        ora.train(train)
        snap = {k: v.clone() for k, v in ora.state_dict().items()}
        with _DecisionMargin() as dm:
            yo = (ofwd or ora)(xo)
        if dm.margin > margin:
            break
    else:
        raise AssertionError("no decision-stable seed found; shrink the case")
    hip.load_state_dict({k: v.float() if v.dtype.is_floating_point else v for k, v in snap.items()})
    hip.to(dev).train(train)
    xh = x.clone().to(dev).requires_grad_(True)
    yh = (fwd or hip)(xh)
    assert rel_err(yh, yo) < tol, ("fwd", rel_err(yh, yo))
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    yh.backward(g.to(dev))
    yo.backward(g.double())
    op = dict(ora.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in op.values() if p.grad is not None)
    errs = {"dx": rel_err(xh.grad, xo.grad)}
    for k, p in hip.named_parameters():
        if op[k].grad is None:
            continue
        a, b = (p.grad if p.grad is not None else torch.zeros_like(p)).double().cpu(), op[k].grad      # (None = all-zero)
        # |err| <= 1e-4 * own scale + 1e-6 * largest gradient of the module: the second term lets
        # analytically-zero gradients pass (conv bias in front of a train-mode BN is pure
        # rounding noise, ~1e-7 * gmax, in the reference's arithmetic as well)
        errs[k] = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-2 * gmax)
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, bad
    ob = dict(ora.named_buffers())
    for k, b in hip.named_buffers():
        if "running" in k:
            assert rel_err(b, ob[k]) < 1e-5, k


FIRE_CASES = [(4, 768, 80, 384, 4, 1, None), (4, 512, 80, 384, 4, 1, "simple"), (2, 64, 16, 64, 8, 32, "simple"),
              (2, 128, 16, 64, 8, 32, "simple"), (3, 256, 48, 192, 5, 9, None), (1, 512, 64, 256, 8, 16, "simple"),
              # 'complex' bypass (pointseg_modules.py:110-112,136-138): 1x1 upsample residual where the
              # width changes, no residual where it does not
              (2, 64, 16, 64, 8, 32, "complex"), (3, 256, 48, 192, 5, 9, "complex"), (2, 128, 16, 64, 8, 32, "complex")]


@pytest.mark.parametrize("case", FIRE_CASES)
@pytest.mark.parametrize("train", [True, False])
def test_fire(dev, case, train):
    from deeplio_amd import nets
    from oracle import model as om
    N, cin, sq, e, H, W, byp = case
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
    compare(nets.Fire(cin, sq, e, e, bypass=byp), om.Fire(cin, sq, e, e, 0.1, byp), x, dev, train)


class _FirePair(nn.Module):
    def __init__(self, mk, a, b, defer):
        super().__init__()
        self.a, self.b, self.defer = mk(*a), mk(*b), defer

    def forward(self, x):
        return self.b(self.a(x, defer=True)) if self.defer else self.b(self.a(x))


PAIR_CASES = [  # N, H, W, first block (bypass-free), second block: the PSEncoder pairs blk1.0->1.1, blk3.2->3.3, blk5.0->5.1
    (2, 8, 32, (64, 16, 64, 64, None), (128, 16, 64, 64, "simple")),
    (1, 4, 32, (384, 64, 256, 256, "simple"), (512, 64, 256, 256, "simple")),
    (2, 4, 8, (512, 80, 384, 384, "simple"), (768, 80, 384, 384, None)),
    (1, 8, 32, (256, 48, 192, 192, "simple"), (384, 48, 192, 192, "simple"))]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_fire_pair_apply_on_load(dev, case):
    """apply-on-load: the first block returns its RAW expand output + (mean, scale, beta); the second
    block's squeeze convolution, bypass residual and squeeze weight gradient activate it while loading.
    Same oracle (two plain Fire blocks), same 1e-4 bars as test_fire; and bit-for-bit the same forward
    values as the materialising path wherever both run the same kernels."""
    from deeplio_amd import nets
    from oracle import model as om
    N, H, W, a, b = case
    x = torch.randn(N, a[0], H, W, generator=torch.Generator().manual_seed(1))
    hmk = lambda ci, sq, e1, e3, byp: nets.Fire(ci, sq, e1, e3, bypass=byp)
    omk = lambda ci, sq, e1, e3, byp: om.Fire(ci, sq, e1, e3, 0.1, byp)
    compare(_FirePair(hmk, a, b, True), _FirePair(omk, a, b, False), x, dev, True)


CBR_CASES = [  # N, cin, cout, k, stride, pad, H, W, pre_relu, bias
    (1, 40, 72, 3, 1, 1, 33, 33, True, True), (2, 72, 64, 3, 1, 1, 17, 17, True, True),
    (2, 10, 64, (3, 5), (1, 2), (1, 2), 16, 64, False, True), (2, 6, 64, (5, 7), (1, 2), (2, 3), 16, 64, False, False),
    (2, 64, 128, (3, 5), (1, 2), (1, 2), 16, 32, False, False), (2, 256, 512, 3, 2, 1, 8, 16, False, False),
    (2, 64, 128, 1, (1, 2), 0, 8, 32, False, False), (1, 24, 24, 3, 1, 1, 64, 65, True, True),
]


class _OraCBR(nn.Module):
    def __init__(self, cin, cout, k, s, p, pre_relu, bias):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, s, p, bias=bias)
        self.bn = nn.BatchNorm2d(cout)
        self.pre = pre_relu

    def forward(self, x):
        return self.bn(F.relu(self.conv(x))) if self.pre else F.relu(self.bn(self.conv(x)))


class _HipCBR(_OraCBR):
    def forward(self, x):
        from deeplio_amd import nets
        return nets._cbr(x, self.conv, self.bn, self.training, pre_relu=self.pre, post_relu=not self.pre)


@pytest.mark.parametrize("case", CBR_CASES)
def test_conv_bn_act(dev, case):
    N, cin, cout, k, s, p, H, W, pre, bias = case
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(2))
    compare(_HipCBR(cin, cout, k, s, p, pre, bias), _OraCBR(cin, cout, k, s, p, pre, bias), x, dev, True)
    compare(_HipCBR(cin, cout, k, s, p, pre, bias), _OraCBR(cin, cout, k, s, p, pre, bias), x, dev, False)


@pytest.mark.parametrize("pool", [None, (3, (1, 2), (1, 1)), (3, (2, 2), (1, 1))])
def test_se_layer_with_fused_pool(dev, pool):
    from deeplio_amd import nets
    from oracle import model as om
    x = F.relu(torch.randn(2, 32, 8, 16, generator=torch.Generator().manual_seed(3)))
    hip, ora = nets.SELayer(32, 2), om.SELayer(32, 2)
    ofwd = (lambda t: ora(t)) if pool is None else (lambda t: F.max_pool2d(ora(t), pool[0], pool[1], pool[2]))
    compare(hip, ora, x, dev, True, fwd=lambda t: hip(t, pool), ofwd=ofwd)


def test_basic_block_and_encoders(dev):
    from deeplio_amd import nets
    from oracle import model as om
    x = torch.randn(2, 64, 8, 32, generator=torch.Generator().manual_seed(4))
    down_h = nn.Sequential(nn.Conv2d(64, 128, 1, (1, 2), bias=False), nn.BatchNorm2d(128))
    down_o = nn.Sequential(nn.Conv2d(64, 128, 1, (1, 2), bias=False), nn.BatchNorm2d(128))
    compare(nets.BasicBlock(64, 128, (1, 2), down_h), om.BasicBlock(64, 128, (1, 2), down_o), x, dev)
    compare(nets.BasicBlock(64, 64), om.BasicBlock(64, 64), x, dev)
    _ctx(dev)
    xe = torch.randn(2, 10, 16, 64, generator=torch.Generator().manual_seed(5))
    # whole encoder: 40 layers deep with 16-sample batch statistics at the end -> forward only at
    # 3e-4 vs fp64 (torch's own fp32 forward is 1.5e-4 from fp64 here); gradients are covered per layer
    hip, ora = nets.PSEncoder((10, 16, 64), {'bypass': 'simple'}), om.PSEncoder(10, 'simple')
    gc.fill_state(ora, 77)
    hip.load_state_dict(ora.state_dict())
    yh = hip.to(dev).train()(xe.to(dev))
    yo = ora.double().train()(xe.double())
    assert rel_err(yh, yo) < 3e-4


def test_psencoder_complex_bypass(dev):
    """cfg['bypass'] = 'complex' through PSEncoder (pointseg_net.py:13,24-55): state_dict keys incl. the
    upsample convolutions match the oracle's, forward matches the fp64 oracle"""
    from deeplio_amd import nets
    from oracle import model as om
    _ctx(dev)
    xe = torch.randn(2, 10, 16, 64, generator=torch.Generator().manual_seed(5))
    hip, ora = nets.PSEncoder((10, 16, 64), {'bypass': 'complex'}), om.PSEncoder(10, 'complex')
    assert set(hip.state_dict()) == set(ora.state_dict())
    assert any("upsample" in k for k in hip.state_dict())
    gc.fill_state(ora, 77)
    hip.load_state_dict(ora.state_dict())
    yh = hip.to(dev).train()(xe.to(dev))
    y32 = ora.train()(xe)                                       # the reference's own arithmetic (torch fp32 CPU)
    yo = ora.double().train()(xe.double())
    # 40 layers with 16-sample batch statistics at the end: fp32 itself is ~2e-4 from fp64 here; the HIP
    # result has to be as close to fp64 as torch's fp32 is (factor 2 + the 1e-4 of north_star)
    e_hip, e_t32 = rel_err(yh, yo), rel_err(y32, yo)
    assert e_hip < 1e-4 + 2 * e_t32, (e_hip, e_t32)


@pytest.mark.parametrize("mode", ["lstm", "gru"])
@pytest.mark.parametrize("H,L,bidir", [(32, 2, True), (8, 2, True), (16, 1, False)])
def test_imu_rnn_with_state_carry(dev, mode, H, L, bidir):
    from deeplio_amd import nets
    from oracle import model as om
    cfg = _ctx(dev, seq=3)
    rc = {'type': mode, 'input-size': 6, 'hidden-size': H, 'num-layers': L, 'bidirectional': bidir, 'dropout': 0.}
    x = torch.rand(3, 3, 5, 6, generator=torch.Generator().manual_seed(6))
    compare(nets.ImufeatRNN0(rc), om.ImufeatRNN0(rc, om.Ctx(cfg)), x, dev)


def test_odom_rnn_fc_softfusion_imufc(dev):
    from deeplio_amd import nets
    from oracle import model as om
    cfg = _ctx(dev, seq=3)
    ctx = om.Ctx(cfg)
    x = torch.randn(3, 3, 48, generator=torch.Generator().manual_seed(7))
    rc = {'type': 'lstm', 'hidden-size': 1024, 'num-layers': 2, 'bidirectional': True, 'dropout': 0.}
    compare(nets.OdomFeatRNN(48, rc), om.OdomFeatRNN(48, rc, ctx), x, dev)
    compare(nets.OdomFeatFC(48, {'dropout': 0.}), om.OdomFeatFC(48, {'dropout': 0.}, ctx), x, dev)
    shapes = [[1, 3, 32], [1, 3, 16]]
    hip, ora = nets.DeepLIOFusionSoft(shapes, {}), om.DeepLIOFusionSoft(shapes, {}, ctx)
    compare(hip, ora, x, dev, fwd=lambda t: hip([t[..., :32], t[..., 32:]]),
            ofwd=lambda t: ora([t[..., :32], t[..., 32:]]))
    fc = {'input-size': 6, 'hidden-size': [16, 32, 8], 'dropout': 0.}
    xi = torch.rand(3, 3, 7, 6, generator=torch.Generator().manual_seed(8))
    compare(nets.ImuFeatFC(fc), om.ImuFeatFC(fc, ctx), xi, dev)


@pytest.mark.parametrize("B,S,H,bidir", [(8, 2, 1024, True), (8, 5, 512, True), (3, 3, 256, True), (4, 2, 256, False), (1, 1, 256, True)])
def test_odom_lstm_one_launch_sequence_per_layer(dev, B, S, H, bidir):
    """OdomFeatRNN (odom_feat_nets.py:48-86: nn.LSTM(256 -> H, 2 layers, bidirectional) over the S axis, forward half kept) on
    functional.LstmStackFn / csrc/lstm_stream.hip -- both directions and all steps of a layer in one call, the weight-streaming
    products cut into K slices (fp32 MFMA) / N slabs with the partial sums joined by the consumer -- against nn.LSTM in fp64:
    output, input gradient and all 16 parameter gradients (the top layer's reverse direction: exactly zero); the
    headline geometry (B = 8, S = 2, H = 1024), S = 5 (40 rows: three passes of the input projection), a ragged batch, one
    direction, a single step; and against RNNFn's per-direction path on the same weights (1e-6)."""
    from deeplio_amd import functional as Fh
    from deeplio_amd import nets
    from oracle import model as om
    cfg = _ctx(dev, seq=S)
    ctx = om.Ctx(cfg)
    x = torch.randn(B, S, 256, generator=torch.Generator().manual_seed(17))
    rc = {'type': 'lstm', 'hidden-size': H, 'num-layers': 2, 'bidirectional': bidir, 'dropout': 0.}
    hip = nets.OdomFeatRNN(256, rc)
    taken = []
    apply_ = Fh.LstmStackFn.apply
    Fh.LstmStackFn.apply = staticmethod(lambda *a: (taken.append(1), apply_(*a))[1])
    try:
        compare(hip, om.OdomFeatRNN(256, rc, ctx), x, dev, tol=2e-5)
    finally:
        Fh.LstmStackFn.apply = apply_
    assert taken == [1]
    # the same module through the per-direction path
    grads_a = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in hip.named_parameters()}
    xa = x.clone().to(dev).requires_grad_(True)
    ya = hip(xa)
    for p in hip.parameters():
        p.grad = None
    Fh._LSTM_LAYER[0] = False
    try:
        xb = x.clone().to(dev).requires_grad_(True)
        yb = hip(xb)
    finally:
        Fh._LSTM_LAYER[0] = True
    assert rel_err(ya, yb) < 1e-6
    g = torch.randn(ya.shape, generator=torch.Generator().manual_seed(5)).to(dev)
    yb.backward(g)
    gmax = max(float(v.abs().max()) for v in grads_a.values())
    for k, p in hip.named_parameters():
        assert float((p.grad - grads_a[k]).abs().max()) <= 2e-6 * gmax + 1e-6 * float(grads_a[k].abs().max()), k
    if bidir:
        # Q3: the discarded half of the top layer gets no gradient -- the streamed path does not even run that direction
        assert dict(hip.named_parameters())["rnn.weight_ih_l1_reverse"].grad.abs().max() == 0.0     # (per-direction path: zeros)
        assert float(grads_a["rnn.weight_ih_l1_reverse"].abs().max()) == 0.0


def test_stem_pool_apply_on_load(dev):
    """PSEncoder stem + pool1 as one node: pool1 takes the maximum over max(0, BN(raw)) while it loads the stem's raw
    output (dlio_maxpool2d_fwd_aff).  Forward, input-free backward (weight / BatchNorm gradients) and the running
    statistics against the materialising path, which is tested against the oracle elsewhere."""
    from deeplio_amd import nets
    torch.manual_seed(4)
    cfg = {'bypass': 'simple', 'dropout': 0.0, 'classes': ['a', 'b'], 'part': 'encoder'}
    from deeplio_amd import functional as Fh
    res = []
    # fused forward + pool backward folded into the BatchNorm backward (dlio_bn_bwd_pool) | fused forward, separate pool
    # backward | the materialising path
    for fused, fold in ((True, True), (True, False), (False, False)):
        torch.manual_seed(4)
        enc = nets.PSEncoder((10, 16, 128), cfg).to(dev).train()
        nets._STEM_AOL, Fh._POOL_BN_BWD[0] = fused, fold
        x = torch.randn(2, 10, 16, 128, generator=torch.Generator().manual_seed(5)).to(dev)
        try:
            y = next(enc.forward_steps(x))
            y.square().sum().backward()
        finally:
            nets._STEM_AOL, Fh._POOL_BN_BWD[0] = True, True
        c, b = enc.conv1a[0], enc.conv1a[1]
        res.append([y.detach(), c.weight.grad, b.weight.grad, b.bias.grad, b.running_mean.clone(), b.running_var.clone()])
    for other in res[:2]:
        for a, r in zip(other, res[2]):
            assert rel_err(a, r.double()) < 2e-5


# ------------------------------------------------------------------------------ lazy pool gradients (functional._LAZY)
def _fire_se_pool(dev, bypass_first=False, sh=1, seed=5):
    """Fire -> Fire(bypass) -> SELayer -> MaxPool2d(3, (sh, 2), 1) as PSEncoder wires a stage (pointseg_net.py:27-46), at a
    geometry the cooperative BatchNorm kernels take (N = 2, 32 x 256 planes)"""
    from deeplio_amd import nets
    torch.manual_seed(seed)
    f0 = nets.Fire(32, 16, 16, 16, bypass="simple" if bypass_first else None).to(dev)
    f1 = nets.Fire(32, 16, 16, 16, bypass="simple").to(dev)
    se = nets.SELayer(32, reduction=2).to(dev)
    for m in (f0, f1, se):
        m.train()
    x = torch.randn(2, 32, 32, 256, device=dev)
    return f0, f1, se, (3, (sh, 2), (1, 1)), x


def _run_stage(f0, f1, se, pool, x, poolfuse=True, lazy=True, hook=None):
    from deeplio_amd import functional as Fh
    for m in (f0, f1, se):
        m.zero_grad()
    Fh.lazy_clear()
    old = (Fh._POOL_FUSE[0], Fh._LAZY_POOL[0])
    Fh._POOL_FUSE[0], Fh._LAZY_POOL[0] = poolfuse, lazy
    try:
        xi = x.clone().requires_grad_(True)
        a = f0(xi, defer=f0.can_defer())
        r = f1(a, want_gap=True, pool=pool)
        if hook is not None:
            hook(r[0])
        y = se(r[0], pool, r[1], getattr(r[0], "_dlio_pooled", None) if len(r) == 3 else None)
        w = torch.linspace(-1, 1, y.numel(), device=y.device).view_as(y)
        (y * w).sum().backward()
        grads = [xi.grad.clone()] + [p.grad.clone() for m in (f0, f1, se) for p in m.parameters()]
        return y.detach().clone(), grads, len(r) == 3
    finally:
        Fh._POOL_FUSE[0], Fh._LAZY_POOL[0] = old


@pytest.mark.parametrize("sh", [1, 2])
def test_fire_stage_pooling_its_own_output_equals_the_unfused_stage(dev, sh):
    """a PSEncoder stage end (Fire -> Fire(bypass) -> SELayer -> MaxPool, pointseg_net.py:27-46) three ways: the block pools
    while it applies its BatchNorm and never writes its output (poolfuse; its gradient arrives as a lazy entry formed from
    the pooled gradient), lazy pool gradients over the written output, and everything materialised: same forward values bit
    for bit, gradients to summation order"""
    from deeplio_amd import functional as Fh
    f0, f1, se, pool, x = _fire_se_pool(dev, sh=sh)
    y_a, g_a, fused_a = _run_stage(f0, f1, se, pool, x, poolfuse=True, lazy=True)
    y_b, g_b, fused_b = _run_stage(f0, f1, se, pool, x, poolfuse=False, lazy=True)
    y_c, g_c, fused_c = _run_stage(f0, f1, se, pool, x, poolfuse=False, lazy=False)
    assert fused_a and not fused_b and not fused_c and not Fh._LAZY
    assert torch.equal(y_a, y_b) and torch.equal(y_b, y_c)
    for ga, gb, gc_ in zip(g_a, g_b, g_c):
        assert rel_err(ga, gc_) < 5e-6 and rel_err(gb, gc_) < 5e-6


def test_lazy_pool_gradient_refuses_what_would_read_the_unwritten_tensor(dev):
    """the hazards of handing autograd an unwritten gradient: retain_grad / a tensor hook on the block output
    -> the SELayer + pool node writes the gradient itself (same numbers); a pre-pooled input with a hook is refused at
    forward; a second consumer of the block output -> the Fire block raises when its gradient arrives altered; the
    gradient of the block output taken directly -> the pass raises at its end; nothing stays behind in the registry"""
    from deeplio_amd import functional as Fh
    f0, f1, se, pool, x = _fire_se_pool(dev)
    y_ref, g_ref, _ = _run_stage(f0, f1, se, pool, x, poolfuse=False, lazy=False)
    # retain_grad on the written output: not lazy, same gradients, and the retained gradient is the real one
    kept = []
    y1, g1, _ = _run_stage(f0, f1, se, pool, x, poolfuse=False, lazy=True, hook=lambda t: (t.retain_grad(), kept.append(t)))
    assert torch.equal(y1, y_ref) and all(rel_err(a, b) < 5e-6 for a, b in zip(g1, g_ref)) and not Fh._LAZY
    assert kept[0].grad is not None and torch.isfinite(kept[0].grad).all() and float(kept[0].grad.abs().max()) > 0
    # a hook on a pre-pooled output: refused where it is consumed
    with pytest.raises(ValueError):
        _run_stage(f0, f1, se, pool, x, poolfuse=True, lazy=True, hook=lambda t: t.register_hook(lambda g: g))
    Fh.lazy_clear()
    # second consumer of the block output
    Fh._POOL_FUSE[0] = False
    try:
        xi = x.clone().requires_grad_(True)
        out, gap = f1(f0(xi, defer=f0.can_defer()), want_gap=True, pool=pool)
        y = se(out, pool, gap)
        with pytest.raises(RuntimeError, match="second consumer|altered"):
            (y.sum() + out.sum()).backward()
        assert not Fh._LAZY
        # the gradient of the block output itself: nobody consumes the entry -> the pass raises at its end
        xi = x.clone().requires_grad_(True)
        out, gap = f1(f0(xi, defer=f0.can_defer()), want_gap=True, pool=pool)
        y = se(out, pool, gap)
        with pytest.raises(RuntimeError, match="not consumed"):
            torch.autograd.grad(y.sum(), out)
        assert not Fh._LAZY
    finally:
        Fh._POOL_FUSE[0] = True
    # ... and the next ordinary step is unaffected
    y2, g2, _ = _run_stage(f0, f1, se, pool, x, poolfuse=True, lazy=True)
    assert torch.equal(y2, y_ref) and all(rel_err(a, b) < 5e-6 for a, b in zip(g2, g_ref))


def test_lazy_pool_gradient_is_materialised_where_nobody_can_route_it(dev):
    """lazy_materialize: a Fire block with a bypass whose input does NOT come from a block that takes a lazy entry (here: a
    plain tensor) completes the gradient itself -- written output and pre-pooled output alike"""
    from deeplio_amd import functional as Fh
    from deeplio_amd import nets
    torch.manual_seed(9)
    f1 = nets.Fire(32, 16, 16, 16, bypass="simple").to(dev).train()
    se = nets.SELayer(32, reduction=2).to(dev).train()
    pool = (3, (1, 2), (1, 1))
    x = torch.randn(2, 32, 32, 256, device=dev)

    def run(poolfuse, lazy):
        Fh.lazy_clear()
        for m in (f1, se):
            m.zero_grad()
        old = (Fh._POOL_FUSE[0], Fh._LAZY_POOL[0])
        Fh._POOL_FUSE[0], Fh._LAZY_POOL[0] = poolfuse, lazy
        try:
            xi = x.clone().requires_grad_(True)
            r = f1(xi, want_gap=True, pool=pool)
            y = se(r[0], pool, r[1], getattr(r[0], "_dlio_pooled", None) if len(r) == 3 else None)
            (y * y).sum().backward()
            return [xi.grad.clone()] + [p.grad.clone() for m in (f1, se) for p in m.parameters()], len(r) == 3
        finally:
            Fh._POOL_FUSE[0], Fh._LAZY_POOL[0] = old
    ref, _ = run(False, False)
    for pf in (False, True):
        got, fused = run(pf, True)
        assert fused == pf and not Fh._LAZY
        assert all(rel_err(a, b) < 5e-6 for a, b in zip(got, ref))


def test_assign_streams_puts_the_heavy_streams_on_queues_of_their_own(dev):
    """functional.assign_streams: whatever streams the process created before (here: three more), the current stream, the
    second encoder's and the two weight-gradient companions end up on four different hardware queues (two of them on one
    queue cost 2 ms of a 19 ms step: the runtime hands a new stream the least used of its 4 queues); the probe itself
    (dlio_streams_share_queue) says 'shared' for a stream and itself and for the light streams and their partners"""
    from deeplio_amd import functional as Fh, ops
    if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 4:
        pytest.skip("fewer than four hardware queues configured")
    saved, assigned = dict(Fh._AUX), set(Fh._ASSIGNED)
    try:
        extra = [torch.cuda.Stream(device=dev) for _ in range(3)]
        for s in extra:
            with torch.cuda.stream(s):
                torch.zeros(1, device=dev)
        t = Fh.assign_streams(dev, force=True)
        main = torch.cuda.current_stream(dev)
        assert ops.streams_share_queue(main, main)
        enc2 = t["encoder2"]
        heavy = [main, enc2, t["wgrad@%x" % main.cuda_stream], t["wgrad@%x" % enc2.cuda_stream]]
        for i in range(4):
            for j in range(i + 1, 4):
                assert not ops.streams_share_queue(heavy[i], heavy[j]), (i, j)
        imu = t["imu"]
        light = [imu, t["rnndir@%x" % imu.cuda_stream], t["rnndir@%x" % main.cuda_stream]]
        assert len({s.cuda_stream for s in heavy + light}) == 7
        assert not ops.streams_share_queue(imu, main) and not ops.streams_share_queue(imu, enc2)
        assert not ops.streams_share_queue(light[1], imu)
        comm = t["comm"]            # the data-parallel exchange: never on an encoder's queue
        assert not ops.streams_share_queue(comm, main) and not ops.streams_share_queue(comm, enc2)
        # a second call leaves the table alone
        assert Fh.assign_streams(dev) is None and Fh._AUX[(dev.index or 0, "encoder2")] is enc2
    finally:
        Fh._AUX.clear(); Fh._AUX.update(saved)
        Fh._ASSIGNED.clear(); Fh._ASSIGNED.update(assigned)


def test_soft_fusion_and_heads_one_launch_each(dev):
    """functional.SoftFusionFn (fusion_nets.py:64-75) and functional.HeadsFn (deeplio_nets.py:84-90: dropout, fc_pos, fc_ori)
    against torch in fp64 -- values, input gradients, all parameter gradients -- at the headline sizes (16 rows, 128 + 128
    features; 1024-wide features read out of a [.., 2048] buffer whose second half gets a zero gradient) and at ragged ones
    (40 rows = three passes, 48 + 80 features, K = 68); the heads' dropout mask equals the one a separate dlio_dropout_fwd
    launch draws at the same Philox position, and the fused nets.DeepLIOFusionSoft equals its unfused path."""
    from deeplio_amd import functional as Fh, ops
    g = torch.Generator().manual_seed(23)
    for R, Fa, Fb in ((16, 128, 128), (40, 48, 80), (3, 8, 500)):
        a, b = torch.randn(R, Fa, generator=g), torch.randn(R, Fb, generator=g)
        F_ = Fa + Fb
        w1, b1 = torch.randn(Fa, F_, generator=g) / F_ ** 0.5, torch.randn(Fa, generator=g)
        w2, b2 = torch.randn(Fb, F_, generator=g) / F_ ** 0.5, torch.randn(Fb, generator=g)
        t64 = [t.double().requires_grad_(True) for t in (a, b, w1, b1, w2, b2)]
        cat = torch.cat(t64[:2], 1)
        s1, s2 = torch.sigmoid(cat @ t64[2].T + t64[3]), torch.sigmoid(cat @ t64[4].T + t64[5])
        ref = torch.cat([t64[0] * s1, t64[1] * s2], 1)
        dy = torch.randn(R, F_, generator=g)
        ref.backward(dy.double())
        th = [t.to(dev).requires_grad_(True) for t in (a, b, w1, b1, w2, b2)]
        out, gate = Fh.SoftFusionFn.apply(*th)
        assert rel_err(out, ref) < 2e-6 and rel_err(gate, torch.cat([s1, s2], 1)) < 2e-6
        out.backward(dy.to(dev))
        for h, r in zip(th, t64):
            assert rel_err(h.grad, r.grad) < 5e-6, (R, Fa, Fb, tuple(h.shape))
    # the second feature as a strided slice of a larger tensor (the IMU net's rnn_out[:, :, -1, :H]): read in place, same values
    big = torch.randn(4, 2, 5, 64, generator=g).to(dev)
    a = torch.randn(4, 2, 32, generator=g).to(dev)
    w1, b1 = (torch.randn(32, 64, generator=g) / 8).to(dev), torch.randn(32, generator=g).to(dev)
    w2, b2 = (torch.randn(32, 64, generator=g) / 8).to(dev), torch.randn(32, generator=g).to(dev)
    sl = big[:, :, -1, :32]
    assert not sl.is_contiguous() and Fh._rows_view(sl)[1] == 5 * 64
    o1, g1 = Fh.SoftFusionFn.apply(a, sl, w1, b1, w2, b2)
    o2, g2 = Fh.SoftFusionFn.apply(a, sl.contiguous(), w1, b1, w2, b2)
    assert torch.equal(o1, o2) and torch.equal(g1, g2) and o1.shape == (4, 2, 64)
    for R, K, ldx, p in ((16, 1024, 2048, 0.25), (40, 68, 68, 0.5), (5, 128, 256, 0.0)):
        x = torch.randn(R, ldx, generator=g)
        wp, bp = torch.randn(3, K, generator=g) / K ** 0.5, torch.randn(3, generator=g)
        wo, bo = torch.randn(3, K, generator=g) / K ** 0.5, torch.randn(3, generator=g)
        th = [t.to(dev).requires_grad_(True) for t in (x, wp, bp, wo, bo)]
        Fh.manual_seed(77)
        pos, ori = Fh.HeadsFn.apply(th[0].view(1, R, ldx), th[1], th[2], th[3], th[4], p, True)
        if p > 0:
            Fh.manual_seed(77)
            _, mask = Fh._dropout_launch(x[:, :K].contiguous().to(dev), p)       # the separate launch: same stream position
            assert Fh.dropout_offset() == (R * K + 3) // 4
            keep = mask.cpu().double() / (1.0 - p)
        else:
            keep = torch.ones(R, K, dtype=torch.float64)
        t64 = [t.double().requires_grad_(True) for t in (x, wp, bp, wo, bo)]
        y = t64[0][:, :K] * keep
        rpos, rori = y @ t64[1].T + t64[2], y @ t64[3].T + t64[4]
        assert rel_err(pos[0], rpos) < 2e-6 and rel_err(ori[0], rori) < 2e-6
        d1, d2 = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
        (rpos * d1.double()).sum().add((rori * d2.double()).sum()).backward()
        (pos[0] * d1.to(dev)).sum().add((ori[0] * d2.to(dev)).sum()).backward()
        for h, r in zip(th, t64):
            assert rel_err(h.grad, r.grad) < 5e-6, (R, K, tuple(h.shape))
        if ldx > K:
            assert float(th[0].grad[:, K:].abs().max()) == 0.0
