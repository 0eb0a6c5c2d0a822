"""-m gpu tests of the mixed-precision path (BASELINE configs[4]: bf16 activation / gradient storage,
fp32 master weights, accumulation, statistics and loss; deeplio_amd/mixed.py, csrc/conv_bf16.hip,
csrc/mixed_bf16.hip).

There is no bf16 reference (the reference is fp32 only).  Two kinds of checks:
  * kernel level: every bf16 entry point against fp64 torch arithmetic on the SAME bf16-rounded operands.
    A bf16 output may differ from the correctly rounded value by one rounding: |err| <= 2^-8 of the
    element (+ fp32 accumulation noise), asserted as BF16_TOL = 5e-3 of the tensor scale; fp32 outputs
    (weight gradients, statistics) at 1e-4.
  * model level: the bf16 model against the fp32 ORACLE with the same weights: tolerance 3e-2 of the
    output scale ("fp32 oracle with a stated bf16 tolerance"; 40 layers of 2^-9 relative
    roundings, re-normalised by BatchNorm), the fp32 HIP model at the same seeds as the fp32 golden, and
    training behaviour (loss trajectory next to the fp32 one)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_common as gc  # noqa: E402
from conftest import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
BF16_TOL = 5e-3
MODEL_TOL = 3e-2


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _r(shape, seed, scale=1.0):
    """bf16-representable fp32 values"""
    return (torch.randn(shape, generator=_g(seed)) * scale).bfloat16().float()


def test_cast_round_trip(dev):
    from deeplio_amd import mixed
    x = torch.randn(3, 5, 8, 16, generator=_g(0))
    y = mixed.cast(x.to(dev), True)
    assert y.dtype == torch.bfloat16 and torch.equal(y.cpu(), x.bfloat16())          # round to nearest even
    z = mixed.cast(y, False)
    assert z.dtype == torch.float32 and torch.equal(z.cpu(), x.bfloat16().float())


@pytest.mark.parametrize("case", [  # N, Cin, Cout, k, H, W, residual
    (2, 16, 64, 3, 8, 64, False), (1, 48, 192, 3, 12, 128, True), (3, 80, 384, 3, 4, 32, False),
    (2, 64, 16, 1, 8, 64, False), (2, 512, 64, 1, 8, 32, True), (1, 48, 192, 1, 16, 128, False),
    (2, 40, 20, 3, 5, 24, True), (2, 24, 40, 1, 6, 20, True)])
@pytest.mark.parametrize("mode", [0, 1])
def test_conv_bf16_forward_and_data_gradient(dev, case, mode):
    """mode 0: y = conv(x, w) + bias (+ residual); mode 1: the data gradient dx = conv_transpose(dy, w)
    (+ residual) through the tap-reversed, transposed layout -- against fp64 on the bf16-rounded operands"""
    from deeplio_amd import mixed, ops
    N, Cin, Cout, k, H, W, res = case
    pad = k // 2
    w = torch.randn(Cout, Cin, k, k, generator=_g(1)) / (Cin * k * k) ** 0.5
    wr = w.bfloat16().double()
    if mode == 0:
        x = _r((N, Cin, H, W), 2)
        b = torch.randn(Cout, generator=_g(3)) * 0.1
        r = _r((N, Cout, H, W), 4) if res else None
        ref = F.conv2d(x.double(), wr, b.double(), 1, pad) + (r.double() if res else 0.)
        d = ops.conv_desc(N, Cin, H, W, Cout, k, k, 1, 1, pad, pad, res_ctot=Cout)
        wt = torch.empty((mixed.lib.dlio_conv_bf16_prep_elems(Cout, Cin, k * k, 0) + 1) // 2, device=dev)
        mixed.check(mixed.lib.dlio_conv_bf16_prep(mixed._ptr(w.to(dev)), mixed._ptr(wt), Cout, Cin, k * k, 0, None), "prep")
        y = torch.empty(N, Cout, H, W, dtype=torch.bfloat16, device=dev)
        mixed.conv_fwd(x.to(dev).bfloat16(), wt, b.to(dev), y, d, residual=r.to(dev).bfloat16() if res else None)
    else:
        dy = _r((N, Cout, H, W), 5)
        r = _r((N, Cin, H, W), 6) if res else None
        ref = F.conv_transpose2d(dy.double(), wr, None, 1, pad) + (r.double() if res else 0.)
        d = ops.conv_desc(N, Cout, H, W, Cin, k, k, 1, 1, k - 1 - pad, k - 1 - pad, OH=H, OW=W, res_ctot=Cin)
        wt = torch.empty((mixed.lib.dlio_conv_bf16_prep_elems(Cout, Cin, k * k, 1) + 1) // 2, device=dev)
        mixed.check(mixed.lib.dlio_conv_bf16_prep(mixed._ptr(w.to(dev)), mixed._ptr(wt), Cout, Cin, k * k, 1, None), "prep")
        y = torch.empty(N, Cin, H, W, dtype=torch.bfloat16, device=dev)
        mixed.conv_fwd(dy.to(dev).bfloat16(), wt, None, y, d, residual=r.to(dev).bfloat16() if res else None)
    assert rel_err(y.float(), ref) < BF16_TOL, rel_err(y.float(), ref)
    # and exactly one rounding away from the fp64 value: never further than 2^-8 of the element (+ accumulation noise)
    err = (y.float().cpu().double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -8 + 1e-4 * float(ref.abs().max())).all())


@pytest.mark.parametrize("case", [(2, 16, 64, 3, 8, 64), (1, 48, 192, 3, 12, 128), (3, 80, 384, 3, 4, 32),
                                  (2, 64, 16, 1, 8, 64), (2, 512, 64, 1, 8, 32), (1, 48, 192, 1, 16, 128),
                                  (4, 128, 16, 1, 16, 64), (2, 16, 64, 1, 8, 32)])
def test_conv_wgrad_bf16(dev, case):
    """dW (fp32) from bf16 x and dy: fp32 accumulation of exact bf16 products -> 1e-4 against fp64, with and
    without accumulation into an existing gradient"""
    from deeplio_amd import mixed, ops
    N, Cin, Cout, k, H, W = case
    pad = k // 2
    x, dy = _r((N, Cin, H, W), 7), _r((N, Cout, H, W), 8)
    wq = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), wq, None, 1, pad) * dy.double()).sum().backward()
    d = ops.conv_desc(N, Cin, H, W, Cout, k, k, 1, 1, pad, pad)
    dw = torch.empty(Cout, Cin, k, k, device=dev)
    mixed.conv_wgrad(x.to(dev).bfloat16(), dy.to(dev).bfloat16(), dw, d)
    assert rel_err(dw, wq.grad) < 1e-4, rel_err(dw, wq.grad)
    base = torch.randn(Cout, Cin, k, k, generator=_g(9))
    dw2 = base.clone().to(dev)
    mixed.conv_wgrad(x.to(dev).bfloat16(), dy.to(dev).bfloat16(), dw2, d, accumulate=True)
    assert rel_err(dw2, wq.grad + base.double()) < 1e-4


@pytest.mark.parametrize("shape", [(4, 24, 8, 64), (2, 64, 16, 32), (16, 8, 4, 16), (3, 40, 2, 8),
                                   (4, 9, 64, 128), (32, 5, 64, 256), (2, 300, 64, 512)])
@pytest.mark.parametrize("res,gap", [(False, False), (True, True)])
def test_batchnorm_bf16_train_eval_backward(dev, shape, res, gap):
    """(the 64 x 128 ... 64 x 512 planes run on the cooperative one-launch kernels of csrc/bn_small.hip: one, two and four
    workgroups per plane, 300 channels x 2 images = several trips of the persistent grid)"""
    from deeplio_amd import mixed, ops
    N, C, H, W = shape
    HW = H * W
    mixed._BN_COOP16[0] = H * W >= 8192            # (off in the product path: slower there; see mixed._BN_COOP16)
    x = _r(shape, 10, 2.0) + 0.5
    x = x.bfloat16().float()
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=_g(11)), 0.1 * torch.randn(C, generator=_g(12))
    r = _r(shape, 13) if res else None
    rm, rv = torch.zeros(C), torch.ones(C)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = rm.double(), rv.double()
    ref = F.relu(F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5)) + (r.double() if res else 0.)
    y = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    gp = torch.empty(N, C, device=dev) if gap else None
    rmh, rvh = rm.to(dev), rv.to(dev)
    prm = mixed.bn_apply(x.to(dev).bfloat16(), C, 0, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rmh, rvh, y, C, 0, N, C, HW,
                         True, r.to(dev).bfloat16() if res else None, C, 0, gp, C, 0)
    assert rel_err(y.float(), ref) < BF16_TOL
    mean = x.double().mean((0, 2, 3))
    var = x.double().var((0, 2, 3), unbiased=False)
    assert rel_err(prm[0], mean) < 1e-5 and rel_err(prm[1], 1 / (var + 1e-5).sqrt()) < 1e-5
    assert rel_err(rmh, rmd) < 1e-5 and rel_err(rvh, rvd) < 1e-5                    # running statistics updated
    if gap:
        assert rel_err(gp, y.float().double().mean((2, 3))) < 1e-6                   # averages of the STORED output
    # backward
    dy = _r(shape, 14)
    ref.backward(dy.double())
    dx = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    mixed.bn_bwd(dy.to(dev).bfloat16(), C, 0, x.to(dev).bfloat16(), C, 0, prm, beta.to(dev), dx, C, 0, N, C, HW, True, True,
                 dg, db)
    assert rel_err(dx.float(), xd.grad) < BF16_TOL
    assert rel_err(dg, gd.grad) < 1e-4 and rel_err(db, bd.grad) < 1e-4
    # eval mode: running statistics
    xe = x.double()
    refe = F.relu(F.batch_norm(xe, rmd, rvd, gamma.double(), beta.double(), False, 0.1, 1e-5))
    ye = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    mixed.bn_apply(x.to(dev).bfloat16(), C, 0, gamma.to(dev), beta.to(dev), 1e-5, 0.1, None, None, ye, C, 0, N, C, HW, True,
                   eval_prm=ops.bn_eval_params(rmh, rvh, gamma.to(dev), 1e-5))
    assert rel_err(ye.float(), refe) < BF16_TOL
    mixed._BN_COOP16[0] = False


@pytest.mark.parametrize("shape,sh", [((2, 6, 8, 32), 1), ((3, 4, 8, 64), 2), ((1, 16, 16, 16), 2), ((2, 3, 4, 48), 1)])
@pytest.mark.parametrize("scaled", [False, True])
def test_maxpool_bf16_with_se_scale(dev, shape, sh, scaled):
    """values bit-exact (max of the scaled fp32 values, rounded once), first-maximum tie rule (bf16 data
    is full of ties), backward and the scale gradient against autograd through F.max_pool2d"""
    from deeplio_amd import mixed
    N, C, H, W = shape
    x = _r(shape, 15)
    x[0, 0, :, :8] = 0.5                                    # a block of exact ties
    s = torch.rand(N, C, generator=_g(16)) * 0.8 + 0.1 if scaled else None
    xs = x.clone().requires_grad_(True)
    sr = s.clone().requires_grad_(True) if scaled else None
    xin = xs * sr[:, :, None, None] if scaled else xs
    ref = F.max_pool2d(xin, 3, (sh, 2), 1)
    y, idx = mixed.maxpool_fwd(x.to(dev).bfloat16(), 3, sh, 2, 1, 1, x_scale=s.to(dev) if scaled else None)
    assert torch.equal(y.cpu(), ref.detach().bfloat16())
    dy = _r(tuple(ref.shape), 17)
    ref.backward(dy)
    add = torch.randn(N, C, generator=_g(18)) * 0.01
    dx = mixed.maxpool_bwd(dy.to(dev).bfloat16(), idx, shape, 3, sh, 2, 1, 1, x_scale=s.to(dev) if scaled else None,
                           x_add=add.to(dev))
    want = xs.grad + add[:, :, None, None]
    assert rel_err(dx.float(), want) < BF16_TOL
    if scaled:
        ds = mixed.maxpool_bwd_dot(dy.to(dev).bfloat16(), idx, x.to(dev).bfloat16(), 3, sh, 2, 1, 1)
        assert rel_err(ds, sr.grad) < 1e-5


def test_gap_bf16(dev):
    from deeplio_amd import mixed
    x = _r((3, 10, 4, 16), 19)
    g = mixed.gap_fwd(x.to(dev).bfloat16(), 3, 10, 0, 10, 64)
    assert rel_err(g, x.double().mean((2, 3))) < 1e-6
    dg = torch.randn(3, 10, generator=_g(20))
    dx = mixed.gap_bwd(dg.to(dev), (3, 10, 4, 16))
    assert rel_err(dx.float(), (dg / 64)[:, :, None, None].expand(3, 10, 4, 16)) < BF16_TOL


FIRE_CASES = [(2, 64, 16, 64, 8, 32, "simple"), (2, 128, 16, 64, 8, 32, "simple"), (3, 256, 48, 192, 4, 16, None),
              (2, 512, 80, 384, 4, 8, "simple"), (2, 768, 80, 384, 2, 16, None)]


def _l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("case", FIRE_CASES)
@pytest.mark.parametrize("train", [True, False])
def test_fire_bf16_vs_fp64_oracle(dev, case, train):
    """a Fire block on bf16 tensors against the fp64 oracle block with the same (fp32) weights.  Forward:
    2e-2 of the output scale.  Gradients: rounding the activations to bf16 moves ~0.3 % of the ReLU inputs
    across zero, and each such element changes its gradient contribution by 100 % -- a relative L2 error of
    sqrt(0.003) ~ 5 % per block that ANY bf16 execution has (measured: the fp32 kernels on an input
    perturbed by 2^-9 relative noise deviate by 4-6 %, tools/bf16_flip_probe.py); asserted: relative L2 <= 0.12
    and the direction (cosine) >= 0.99 for the input gradient and every weight gradient."""
    from deeplio_amd import nets
    from oracle import model as om
    N, cin, sq, e, H, W, byp = case
    hip, ora = nets.Fire(cin, sq, e, e, bypass=byp), om.Fire(cin, sq, e, e, 0.1, byp)
    gc.fill_state(ora, 77)
    hip.load_state_dict(ora.state_dict())
    hip.to(dev).train(train)
    ora.double().train(train)
    x = _r((N, cin, H, W), 21)
    xo = x.double().requires_grad_(True)
    yo = ora(xo)
    xh = x.to(dev).bfloat16().requires_grad_(True)
    yh = hip(xh)
    assert yh.dtype == torch.bfloat16
    assert rel_err(yh.float(), yo) < 2e-2, rel_err(yh.float(), yo)
    g = _r(tuple(yo.shape), 22)
    yo.backward(g.double())
    yh.backward(g.to(dev).bfloat16())
    assert xh.grad.dtype == torch.bfloat16
    errs = {"dx": (_l2(xh.grad.float(), xo.grad),
                   float(F.cosine_similarity(xh.grad.float().cpu().double().flatten(), xo.grad.flatten(), dim=0)))}
    op = dict(ora.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in op.values())
    for k, p in hip.named_parameters():
        assert p.grad.dtype == torch.float32                   # master-weight gradients stay fp32
        b = op[k].grad
        if float(b.abs().max()) < 1e-5 * gmax:                 # conv bias in front of a train-mode BN: zero
            assert float(p.grad.abs().max()) <= 1e-3 * gmax, k
            continue
        errs[k] = (_l2(p.grad, b), float(F.cosine_similarity(p.grad.double().cpu().flatten(), b.flatten(), dim=0)))
    bad = {k: v for k, v in errs.items() if not (v[0] < 0.12 and v[1] > 0.99)}
    assert not bad, bad
    if train:
        ob = dict(ora.named_buffers())
        for k, b in hip.named_buffers():
            if "running" in k:
                assert rel_err(b, ob[k]) < 2e-2, k


GEOM = dict(B=2, S=2, C=5, H=16, W=1024, T=7)          # smallest width the bf16 pools take (W/64 % 16 == 0)


def _cfg(precision, seq=2, geodesic=False):
    from deeplio_amd.config import make_config
    ov = dict(gc.NO_DROP)
    ov.update(gc.SMALL_RNN)
    ov['lidar-feat-pointseg/precision'] = precision
    cfg = make_config(seq=seq, overrides=ov)
    if geodesic:
        cfg['losses']['rotation'] = 'geodesic'
    return cfg


def _hip_model(cfg, dev, B):
    from deeplio_amd import losses, misc, nets
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    model = nets.get_model((GEOM['C'], GEOM['H'], GEOM['W']), cfg, dev)
    gc.fill_state(model, seed=1000)
    return model, losses.get_loss_function(cfg, dev)


def _step(model, crit, batch):
    from deeplio_amd.se3 import se3_to_SE3
    xyz, nrm, imu, f2f, f2g = batch
    pt, pw = model([[xyz, nrm], imu])
    pp, pq = se3_to_SE3(pt, pw)
    loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    return pt, pw, loss


@pytest.mark.parametrize("train", [False, True])
def test_model_bf16_forward_vs_fp32_oracle(dev, train):
    """PointSeg + bi-LSTM + soft fusion + odometry bi-LSTM with bf16 encoders against the fp32 CPU oracle
    (same fill_state weights, same batch): pose outputs within MODEL_TOL of their scale; the fp32 HIP model
    on the same inputs is within 1e-4 of the oracle (the fp32 golden at the same seeds); state_dict keys
    and dtypes are unchanged (fp32 master weights)"""
    from oracle import model as om
    g = GEOM
    batch = gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
    dbatch = tuple(t.to(dev) for t in batch)
    cfg16, cfg32 = _cfg('bf16'), _cfg('fp32')
    m16, _ = _hip_model(cfg16, dev, g['B'])
    m32, _ = _hip_model(cfg32, dev, g['B'])
    assert set(m16.state_dict()) == set(m32.state_dict())
    assert all(v.dtype == torch.float32 for v in m16.state_dict().values() if v.dtype.is_floating_point)
    omodel = om.get_model((g['C'], g['H'], g['W']), cfg32)
    gc.fill_state(omodel, seed=1000)
    for m in (m16, m32, omodel):
        m.train(train)
    with torch.no_grad():
        p16, o16 = m16([[dbatch[0], dbatch[1]], dbatch[2]])
        p32, o32 = m32([[dbatch[0], dbatch[1]], dbatch[2]])
        po, oo = omodel([[batch[0], batch[1]], batch[2]])
    assert rel_err(p32, po) < 1e-4 and rel_err(o32, oo) < 1e-4
    e = max(rel_err(p16, po), rel_err(o16, oo))
    print("bf16 model vs fp32 oracle (train=%s): rel err pos %.2e ori %.2e" % (train, rel_err(p16, po), rel_err(o16, oo)))
    assert e < MODEL_TOL, e


def _oracle_step(cfg, g, batch, dtype=torch.float32, autocast=False):
    from oracle import model as om
    from oracle import se3 as ose3
    m = om.get_model((g['C'], g['H'], g['W']), cfg)
    gc.fill_state(m, seed=1000)
    m = m.to(dtype).train()
    crit = om.get_loss_function(cfg).to(dtype)
    xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        pt, pw = m([[xyz, nrm], imu])
    pt, pw = pt.to(dtype), pw.to(dtype)
    pp, pq = ose3.se3_to_SE3(pt, pw)
    loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    loss.backward()
    return float(loss.item()), {k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None}


def test_model_bf16_gradients_vs_autocast_oracle(dev):
    """Whole-model gradients.  The fp32 path's whole-model gradients are only ~1e-2 accurate on these
    random-init geometries (tests/test_gpu_model.py: rounding at 6e-8 is amplified ~1e5x by ReLU / max-pool
    decision flips and the BatchNorm in front of the global average pool), so rounding at 2^-9 saturates the
    error: NO bf16 execution reproduces the fp32 gradient direction of this model at initialisation.  The
    oracle for mixed precision is therefore PyTorch's own: the CPU oracle under torch.autocast(bfloat16).
    Asserted: the HIP bf16 gradients are as close to the exact (fp64) gradients as that execution is
    (median cosine over the encoder conv weights >= its median - 0.1), the parameters behind the encoders
    (fc1, RNNs, heads -- fp32 arithmetic on bf16-perturbed features) agree to 0.95, all gradients are fp32
    and finite."""
    g = GEOM
    batch = gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
    dbatch = tuple(t.to(dev) for t in batch)
    cfg16 = _cfg('bf16')
    model, crit = _hip_model(cfg16, dev, g['B'])
    model.train()
    *_, loss = _step(model, crit, dbatch)
    loss.backward()
    g16 = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}
    assert all(p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()) for p in model.parameters()
               if p.grad is not None)
    l64, g64 = _oracle_step(_cfg('fp32'), g, batch, torch.float64)
    lac, gac = _oracle_step(_cfg('fp32'), g, batch, torch.float32, autocast=True)
    assert set(g16) == set(g64)
    assert abs(float(loss.item()) - l64) <= MODEL_TOL * abs(l64)
    enc = [k for k in g64 if "encoder" in k and k.endswith("weight") and g64[k].dim() == 4]
    cos = lambda a, b: float(F.cosine_similarity(a.flatten(), b.flatten(), dim=0))
    c_hip = np.asarray([cos(g16[k], g64[k]) for k in enc])
    c_ac = np.asarray([cos(gac[k], g64[k]) for k in enc])
    print("encoder conv weight gradients, cosine vs fp64: HIP bf16 median %.3f min %.3f | torch autocast(bf16) oracle "
          "median %.3f min %.3f | loss hip %.5f autocast %.5f fp64 %.5f"
          % (np.median(c_hip), c_hip.min(), np.median(c_ac), c_ac.min(), float(loss.item()), lac, l64))
    assert np.median(c_hip) >= np.median(c_ac) - 0.1
    for k in ("lidar_feat_net.fc1.weight", "odom_feat_net.rnn.weight_ih_l0", "fc_pos.weight", "fc_ori.weight",
              "imu_feat_net.rnn.weight_hh_l0"):
        assert cos(g16[k], g64[k]) > 0.95, (k, cos(g16[k], g64[k]))


def test_model_bf16_training_tracks_fp32(dev):
    """BASELINE configs[4] as a training run at small scale: seq_len 4, geodesic rotation loss, bf16
    encoders, Adam on the fp32 master weights.  The loss of the first step agrees with the fp32 oracle
    inside the bf16 envelope and five optimizer steps track the fp32 HIP run (same seeds: losses within 10 %)
    and decrease the loss."""
    from deeplio_amd.optimizer import create_optimizer
    from oracle import model as om
    from oracle import se3 as ose3
    g = dict(GEOM, S=4, B=2)
    batch = gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
    dbatch = tuple(t.to(dev) for t in batch)
    args = types.SimpleNamespace(lr=1e-3, weight_decay=1e-4, momentum=0.9)
    traj = {}
    for prec in ("bf16", "fp32"):
        cfg = _cfg(prec, seq=4, geodesic=True)
        model, crit = _hip_model(cfg, dev, g['B'])
        model.train()
        opt = create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg, args)
        losses = []
        for it in range(5):
            *_, loss = _step(model, crit, dbatch)
            opt.zero_grad()
            loss.backward()
            if it == 0:
                grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            opt.step()
            losses.append(float(loss.item()))
        traj[prec] = (losses, grads)
    cfg = _cfg('fp32', seq=4, geodesic=True)
    omodel = om.get_model((g['C'], g['H'], g['W']), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.train()
    ocrit = om.get_loss_function(cfg)
    a, b = omodel([[batch[0], batch[1]], batch[2]])
    p2, q2 = ose3.se3_to_SE3(a, b)
    oloss = float(ocrit(a, b, p2[:, 1:3], q2[:, 1:3], batch[3][:, :, 0:3], batch[3][:, :, 3:], batch[4][:, 1:3, 0:3],
                        batch[4][:, 1:3, 3:7]).item())
    l16, g16 = traj["bf16"]
    l32, g32 = traj["fp32"]
    print("losses bf16", l16, "fp32", l32, "oracle step 0", oloss)
    assert abs(l32[0] - oloss) <= 1e-4 * abs(oloss)
    assert abs(l16[0] - oloss) <= MODEL_TOL * abs(oloss)
    assert all(np.isfinite(l16)) and l16[-1] < l16[0]
    for a16, a32 in zip(l16, l32):
        assert abs(a16 - a32) <= 0.1 * abs(a32), (l16, l32)
    assert set(g16) == set(g32)


def test_model_bf16_full_size_train_steps(dev):
    """BASELINE configs[4] at its own geometry -- 64 x 2048 x 5, T = 50, seq_len 4, geodesic rotation loss, bf16 encoders,
    full-size RNNs, Adam -- with per-GPU batch 2 (8 frame pairs, 16 images per encoder; the bench line runs batch 8): the
    launch sizes at which the bf16 kernels take their large-plane branches (statistics splits, pool strips, weight-gradient
    slabs).  Envelope: the poses and the loss of the first step against the fp32 ORACLE on the same weights and batch within
    MODEL_TOL; properties: three optimizer steps next to the fp32 HIP run at the same seeds stay within 10 % of its losses
    (which at this initialisation and lr 1e-3 go up before they come down), every gradient is fp32 and finite, the running statistics of the first step (same weights) are within MODEL_TOL of the fp32 run's."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from deeplio_amd.optimizer import create_optimizer
    from deeplio_amd.se3 import se3_to_SE3
    from oracle import model as om
    from oracle import se3 as ose3
    B, S, C, H, W, T = 2, 4, 5, 64, 2048, 50
    batch = gc.make_batch(2100, B, S, C, H, W, T)
    dbatch = tuple(t.to(dev) for t in batch)

    def cfg_of(prec):
        ov = dict(gc.NO_DROP)
        ov['lidar-feat-pointseg/precision'] = prec
        cfg = make_config(lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn",
                          seq=S, overrides=ov)
        cfg['losses']['rotation'] = 'geodesic'
        return cfg

    def loss_of(crit, pt, pw, pp, pq, f2f, f2g):
        return crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])

    args = types.SimpleNamespace(lr=1e-3, weight_decay=1e-4, momentum=0.9)
    traj = {}
    for prec in ("bf16", "fp32"):
        cfg = cfg_of(prec)
        misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
        model = nets.get_model((C, H, W), cfg, dev)
        gc.fill_state(model, seed=1000)
        crit = losses.get_loss_function(cfg, dev)
        model.train()
        opt = create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg, args)
        ls, first = [], None
        for it in range(3):
            pt, pw = model([[dbatch[0], dbatch[1]], dbatch[2]])
            pp, pq = se3_to_SE3(pt, pw)
            loss = loss_of(crit, pt, pw, pp, pq, dbatch[3], dbatch[4])
            opt.zero_grad()
            loss.backward()
            if it == 0:
                first = (pt.detach().cpu(), pw.detach().cpu())
                stats = {k: v.detach().cpu().double() for k, v in model.state_dict().items() if "running_" in k}
                assert all(p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all())
                           for p in model.parameters() if p.grad is not None)
            opt.step()
            ls.append(float(loss.item()))
        traj[prec] = (ls, first, stats)
        del model, crit, opt
        torch.cuda.empty_cache()
    cfg = cfg_of('fp32')
    omodel = om.get_model((C, H, W), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.train()
    ocrit = om.get_loss_function(cfg)
    with torch.no_grad():
        a, b = omodel([[batch[0], batch[1]], batch[2]])
        p2, q2 = ose3.se3_to_SE3(a, b)
        oloss = float(loss_of(ocrit, a, b, p2, q2, batch[3], batch[4]).item())
    (l16, f16, s16), (l32, f32, s32) = traj["bf16"], traj["fp32"]
    print("full-size configs[4]: losses bf16", l16, "fp32", l32, "oracle step 0", oloss,
          "| poses vs oracle: bf16 %.2e / %.2e, fp32 %.2e / %.2e" % (rel_err(f16[0], a), rel_err(f16[1], b),
                                                                     rel_err(f32[0], a), rel_err(f32[1], b)))
    assert rel_err(f32[0], a) < 1e-4 and rel_err(f32[1], b) < 1e-4 and abs(l32[0] - oloss) <= 1e-4 * abs(oloss)
    assert rel_err(f16[0], a) < MODEL_TOL and rel_err(f16[1], b) < MODEL_TOL
    assert abs(l16[0] - oloss) <= MODEL_TOL * abs(oloss)
    assert all(np.isfinite(l16))
    for a16, a32 in zip(l16, l32):
        assert abs(a16 - a32) <= 0.1 * abs(a32), (l16, l32)
    worst = max((rel_err(s16[k], s32[k]), k) for k in s32)
    print("running statistics bf16 vs fp32 after the first step: worst %.2e (%s)" % worst)
    assert worst[0] < MODEL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 24, 8, 64), (4, 16, 16, 32)])
def test_bf16_batchnorm_synchronised_statistics_equal_the_whole_batch(dev, shape):
    """SyncBN on the bf16 path (BASELINE configs[4] is a DP = 8 config): the per-channel partial sums are all-reduced
    between the statistics and the apply launch (dlio_bn_bf16_apply / _bwd phases 1 / 2, as the fp32 kernels).  Two
    "ranks" are played in one process -- each half of the batch is run once to collect its partials, then again with an
    all-reduce callable that returns the sum: outputs, statistics, running statistics and dx of the halves equal the
    whole batch's, dgamma / dbeta add up to it."""
    from deeplio_amd import mixed, ops
    N, C_, H, W = shape
    HW = H * W
    g0 = _g(21)
    x = torch.randn(N, C_, H, W, generator=g0).to(dev).bfloat16()
    dy = torch.randn(N, C_, H, W, generator=g0).to(dev).bfloat16()
    gamma = (1.0 + 0.2 * torch.randn(C_, generator=g0)).to(dev)
    beta = (0.2 * torch.randn(C_, generator=g0)).to(dev)

    def run(xs, dys, n):
        rm, rv = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
        y = torch.empty_like(xs)
        prm = mixed.bn_apply(xs, C_, 0, gamma, beta, 1e-5, 0.1, rm, rv, y, C_, 0, n, C_, HW, True)
        dx = torch.empty_like(xs)
        dg, db = torch.zeros(C_, device=dev), torch.zeros(C_, device=dev)
        mixed.bn_bwd(dys, C_, 0, xs, C_, 0, prm, beta, dx, C_, 0, n, C_, HW, True, True, dg, db)
        torch.cuda.synchronize()
        return y, prm.clone(), rm, rv, dx, dg, db
    whole = run(x, dy, N)
    h = N // 2
    halves = [(x[:h].contiguous(), dy[:h].contiguous()), (x[h:].contiguous(), dy[h:].contiguous())]
    # pass 1: collect every all-reduce operand of both ranks, in call order (forward, backward)
    seen = [[], []]
    for r, (xs, dys) in enumerate(halves):
        ops.set_sync_bn(lambda t, r=r: seen[r].append(t.clone()), 2)
        try:
            run(xs, dys, h)
        finally:
            ops.set_sync_bn(None, 1)
    assert len(seen[0]) == 2 and len(seen[1]) == 2
    fwd_sum = seen[0][0] + seen[1][0]
    # pass 2: the all-reduce returns the sum over the two ranks (the backward partials depend on the synchronised forward
    # statistics, so they are collected again under them)
    bwd = [None, None]
    for r, (xs, dys) in enumerate(halves):
        calls = []

        def reduce_(t, r=r, calls=calls):
            if not calls:
                t.copy_(fwd_sum)
            else:
                bwd[r] = t.clone()
            calls.append(1)
        ops.set_sync_bn(reduce_, 2)
        try:
            run(xs, dys, h)
        finally:
            ops.set_sync_bn(None, 1)
    bwd_sum = bwd[0] + bwd[1]
    outs = []
    for r, (xs, dys) in enumerate(halves):
        calls = []

        def reduce2(t, calls=calls):
            t.copy_(fwd_sum if not calls else bwd_sum)
            calls.append(1)
        ops.set_sync_bn(reduce2, 2)
        try:
            outs.append(run(xs, dys, h))
        finally:
            ops.set_sync_bn(None, 1)
    y = torch.cat([outs[0][0], outs[1][0]])
    dx = torch.cat([outs[0][4], outs[1][4]])
    # the same statistics up to fp64 summation order -> the same bf16 roundings (a last-bit difference of a mean may flip one)
    assert float((y.float() - whole[0].float()).abs().mean()) < 1e-5 and rel_err(y.float(), whole[0].float().double()) < 1e-2
    for r in range(2):
        assert rel_err(outs[r][1], whole[1].double()) < 1e-6      # mean, invstd, scale
        assert rel_err(outs[r][2], whole[2].double()) < 1e-6 and rel_err(outs[r][3], whole[3].double()) < 1e-6
    assert rel_err(dx.float(), whole[4].float().double()) < 1e-2 and float((dx.float() - whole[4].float()).abs().mean()) < 1e-4
    assert rel_err(outs[0][5] + outs[1][5], whole[5].double()) < 1e-5
    assert rel_err(outs[0][6] + outs[1][6], whole[6].double()) < 1e-5
