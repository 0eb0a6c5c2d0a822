"""-m gpu parity tests of whole models: the HIP path (deeplio_amd.nets via the C-ABI) against
(a) the golden vectors captured from the reference and (b) the oracle on the same seeded
weights and inputs -- eval forward, train-mode forward/backward (every parameter gradient and
BN running statistic), SE(3) chain + loss, and a short Adam trajectory.
Tolerance 1e-4 relative to each tensor's scale (north_star), stated per assert."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_common as gc  # noqa: E402
from conftest import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4
GOLD = os.path.join(HERE, "golden")


def build(name, dev, train):
    from deeplio_amd import losses, misc, nets
    g = gc.MODEL_CASES[name]['geom']
    cfg = gc.case_cfg(name)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=g['B']))
    model = nets.get_model((g['C'], g['H'], g['W']), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train(train)
    crit = losses.get_loss_function(cfg, dev)
    batch = tuple(t.to(dev) for t in gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T']))
    return cfg, model, crit, batch


def build_oracle(name, train):
    from oracle import model as om
    g = gc.MODEL_CASES[name]['geom']
    cfg = gc.case_cfg(name)
    model = om.get_model((g['C'], g['H'], g['W']), cfg)
    gc.fill_state(model, seed=1000)
    model.train(train)
    crit = om.get_loss_function(cfg)
    batch = gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
    return cfg, model, crit, batch


def hip_step_forward(model, crit, batch):
    from deeplio_amd.se3 import se3_to_SE3
    xyz, nrm, imu, gt_f2f, gt_f2g = batch
    pt, pw = model([[xyz, nrm], imu])
    pp, pq = se3_to_SE3(pt, pw)
    loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:],
                gt_f2g[:, 1:3, 0:3], gt_f2g[:, 1:3, 3:7])
    return pt, pw, pp, pq, loss


@pytest.mark.parametrize("name", list(gc.MODEL_CASES))
def test_eval_forward_vs_reference_golden(dev, name):
    gold = np.load(os.path.join(GOLD, "model_%s.npz" % name))
    _, model, _, batch = build(name, dev, train=False)
    with torch.no_grad():
        pos, ori = model([[batch[0], batch[1]], batch[2]])
    assert rel_err(pos, torch.from_numpy(gold['eval_pos'])) < TOL
    assert rel_err(ori, torch.from_numpy(gold['eval_ori'])) < TOL


def _oracle_grads(name, dtype):
    from oracle import se3 as ose3
    _, omodel, ocrit, obatch = build_oracle(name, train=True)
    omodel, ocrit = omodel.to(dtype), ocrit.to(dtype)
    xyz, nrm, imu, gt_f2f, gt_f2g = (t.to(dtype) for t in obatch)
    pt, pw = omodel([[xyz, nrm], imu])
    pp, pq = ose3.se3_to_SE3(pt, pw)
    loss = ocrit(pt, pw, pp[:, 1:3], pq[:, 1:3], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:],
                 gt_f2g[:, 1:3, 0:3], gt_f2g[:, 1:3, 3:7])
    loss.backward()
    grads = {k: p.grad.double() for k, p in omodel.named_parameters() if p.grad is not None}
    grads["criterion.sx"], grads["criterion.sq"] = ocrit.sx.grad.double(), ocrit.sq.grad.double()
    return (pt, pw, pp, pq, loss), grads, omodel


def _l2(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


@pytest.mark.parametrize("name", list(gc.MODEL_CASES))
def test_train_forward_backward_vs_oracle(dev, name):
    """Train-mode forward: strict 1e-4 (of the tensor scale) against the fp32 oracle and the
    reference golden.  Backward: the golden geometries are tiny (16..128 samples per BN channel
    in the last blocks, a BN directly in front of the global average pool), so the reference's
    OWN fp32 gradients are 1e-2 away from the exact (fp64) gradients of the same function
    (measured: torch fp32 vs fp64, median 2.4e-2 on pointseg_lstm_cat) and no fp32
    implementation can agree with another to 1e-4 there.  The criterion is therefore: per
    parameter, relative L2 error against the fp64 oracle <= max(1e-3, 3 x the error of the
    reference arithmetic (torch fp32) against the same fp64 oracle).  Layer-level gradient
    parity at 1e-4 is asserted in test_gpu_modules.py on decision-stable inputs."""
    gold = np.load(os.path.join(GOLD, "model_%s.npz" % name))
    _, model, crit, batch = build(name, dev, train=True)
    pt, pw, pp, pq, loss = hip_step_forward(model, crit, batch)
    loss.backward()
    (opt_, opw, opp, opq, oloss), g32, omodel = _oracle_grads(name, torch.float32)
    _, g64, _ = _oracle_grads(name, torch.float64)

    assert rel_err(pt, opt_) < TOL and rel_err(pw, opw) < TOL
    assert rel_err(pp, opp) < TOL and rel_err(pq, opq) < TOL
    assert rel_err(loss, oloss) < TOL
    if int(gold['has_bwd']):     # and against the reference's own numbers
        assert rel_err(pt, torch.from_numpy(gold['train_pos'])) < TOL
        assert rel_err(loss, torch.from_numpy(gold['loss'])) < TOL
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gmax = max(float(v.abs().max()) for v in g64.values())
    e_hip, e_ref = [], []
    for k, ref in g64.items():
        assert named[k].grad is not None, k
        mine = named[k].grad.detach().double().cpu()
        if float(ref.abs().max()) < 1e-5 * gmax:       # analytically-zero gradients: noise only
            assert float(mine.abs().max()) < 1e-4 * gmax, k
            continue
        e_hip.append(_l2(mine, ref))
        e_ref.append(_l2(g32[k], ref))
    e_hip, e_ref = np.asarray(e_hip), np.asarray(e_ref)
    print("grad rel-L2 vs fp64: hip median %.2e max %.2e | torch-fp32 median %.2e max %.2e"
          % (np.median(e_hip), e_hip.max(), np.median(e_ref), e_ref.max()))
    # both are samples of the same rounding-noise process (ReLU / max-pool decision flips,
    # cancellation in BN backward): compare the distributions, not parameter by parameter
    assert np.median(e_hip) <= max(1e-3, 3.0 * np.median(e_ref)), (np.median(e_hip), np.median(e_ref))
    assert e_hip.max() <= max(2e-2, 10.0 * e_ref.max()), (e_hip.max(), e_ref.max())
    for k, p in model.named_parameters():
        if k not in g64:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    obufs = dict(omodel.named_buffers())
    for k, b in model.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(b, obufs[k]) < 1e-5, k
        elif k.endswith("num_batches_tracked"):
            assert int(b.item()) == int(obufs[k].item()), k


@pytest.mark.parametrize("name", ["pointseg_lstm_cat", "simple1_fc_soft_cfg1"])
def test_adam_trajectory(dev, name):
    """5 Adam steps against the reference golden.  Step 1 (pure forward) is tight (1e-4).  From
    step 2 on the trajectory inherits the fp32 gradient noise described above, amplified by
    Adam's sign-like first updates (update = lr*g/(|g|+eps): a gradient that is rounding noise
    moves its parameter by a full +-lr): on CPU, merely batching the reference's per-sample IMU
    matmuls moves step 5 of the Simple-1 case by 2.5e-3.  Envelope: 2e-3 at step 2, 2e-2 after."""
    from deeplio_amd.optimizer import create_optimizer
    gold = np.load(os.path.join(GOLD, "traj_%s.npz" % name))
    cfg, model, crit, batch = build(name, dev, train=True)
    args = types.SimpleNamespace(lr=1e-3, weight_decay=1e-4, momentum=0.9)
    opt = create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg, args)
    losses = []
    for _ in range(5):
        *_, loss = hip_step_forward(model, crit, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    ref = gold['losses']
    bounds = [1e-4, 2e-3, 2e-2, 2e-2, 2e-2]
    for got, want, tol in zip(losses, ref, bounds):
        assert abs(got - want) <= tol * abs(want), (losses, ref.tolist())


def test_headline_shapes_smoke(dev):
    """BASELINE config 2 architecture at full 64x2048x5 resolution, B=1: finite outputs, one
    optimizer step, loss decreases on a repeated batch (size-independent sanity)."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from deeplio_amd.optimizer import create_optimizer
    cfg = make_config(seq=2)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=1))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    crit = losses.get_loss_function(cfg, dev)
    batch = tuple(t.to(dev) for t in gc.make_batch(7, 1, 2, 5, 64, 2048, 50))
    args = types.SimpleNamespace(lr=1e-3, weight_decay=1e-4, momentum=0.9)
    opt = create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg, args)
    model.train()
    vals = []
    for _ in range(3):
        *_, loss = hip_step_forward(model, crit, batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        vals.append(float(loss.item()))
    assert all(np.isfinite(vals)) and vals[-1] < vals[0], vals


def test_stream_overlap_is_bit_exact(dev):
    """The IMU branch, the normals encoder and all weight-gradient kernels run on side HIP streams.  Every kernel on the path
    is deterministic (fixed-order split reductions, no float atomics), so overlapping them must
    not change a single bit of the outputs, the loss or any gradient -- a race would."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    cfg = make_config(seq=2)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=2))
    batch = tuple(t.to(dev) for t in gc.make_batch(11, 2, 2, 5, 64, 512, 50))
    res = []
    from deeplio_amd import functional as Fh
    wgrad_default = Fh._WGRAD_FORK[0]
    for overlap in (True, False, True):
        model = nets.get_model((5, 64, 512), cfg, dev)
        gc.fill_state(model, seed=1000)
        crit = losses.get_loss_function(cfg, dev)
        model.train()
        for m in model.modules():
            if hasattr(m, "two_streams"):
                m.two_streams = overlap
            if hasattr(m, "side_stream"):
                m.side_stream = overlap
        from deeplio_amd import functional as Fh
        Fh.manual_seed(5)
        Fh.set_wgrad_stream(overlap)     # fork of the weight-gradient kernels
        outs = []
        for _ in range(2):
            model.zero_grad()
            pt, pw, pp, pq, loss = hip_step_forward(model, crit, batch)
            loss.backward()
            torch.cuda.synchronize()
            outs.append([loss.detach().clone(), pt.detach().clone()]
                        + [p.grad.detach().clone() for p in model.parameters() if p.grad is not None])
        res.append(outs)
    Fh.set_wgrad_stream(wgrad_default)
    for other in res[1:]:
        for a_step, b_step in zip(res[0], other):
            assert len(a_step) == len(b_step)
            for a, b in zip(a_step, b_step):
                assert torch.equal(a, b)


def test_dp_tail_bucket_is_final_when_the_hook_fires(dev):
    """Data parallel: the all-reduce of the odometry-net / head gradients (87 % of the flat buffer)
    is started from an autograd hook on the fusion output.  Property that makes this correct: at
    that point, on the hook's stream, flat_grad[tail:] already holds its final value, and nothing
    in the head of the buffer aliases it."""
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep
    cfg = make_config(seq=2)
    ts = TrainStep(cfg, (5, 64, 256), dev, 2)
    batch = tuple(t.to(dev) for t in gc.make_batch(3, 2, 2, 5, 64, 256, 50))
    seen = {}

    class FakeSync:
        world, tail_lo = 2, None

        def set_tail(self, lo):
            self.tail_lo = lo

        def reduce_tail_async(self):
            seen['tail'] = ts.optimizer.grad[self.tail_lo:].clone()      # on the hook's stream
            seen['calls'] = seen.get('calls', 0) + 1

        def all_reduce_grads(self):
            seen['final'] = ts.optimizer.grad.clone()

    sync = FakeSync()
    ts.set_grad_sync(sync)
    lo = sync.tail_lo
    n = ts.optimizer.grad.numel()
    assert lo is not None and 0 < lo < n and lo % 16 == 0
    assert (n - lo) > 0.8 * n                      # the odometry LSTM dominates the parameter count
    first_odom = next(ts.model.odom_feat_net.parameters())
    assert first_odom.data_ptr() == ts.optimizer.flat.data_ptr() + 4 * lo
    ts.step(*batch)
    torch.cuda.synchronize()
    assert seen['calls'] == 1
    assert torch.equal(seen['tail'], seen['final'][lo:])
    assert float(seen['final'][lo:].abs().max()) > 0 and float(seen['final'][:lo].abs().max()) > 0


@pytest.mark.parametrize("name", [n for n in gc.MODEL_CASES])
def test_gradient_checksums_vs_reference_golden(dev, name):
    """The goldens hold the REFERENCE's own per-parameter gradient checksums (`grad_sums`: sum, abs-sum,
    first / last four elements of every parameter gradient of its fp32 CPU step).  Compare the HIP
    gradients' abs-sums with them and put both next to the fp64 oracle, so that the envelope of
    test_train_forward_backward_vs_oracle is evidence: HIP must be as close to the exact gradients as
    the reference's own arithmetic is (factor 3); the direct HIP-vs-reference difference is printed and
    bounded by the two distances from fp64 (both are fp32 roundings of the same function)."""
    gold = np.load(os.path.join(GOLD, "model_%s.npz" % name))
    if not int(gold['has_bwd']):
        pytest.skip("reference backward undefined for this configuration (SURVEY Q2)")
    _, model, crit, batch = build(name, dev, train=True)
    *_, loss = hip_step_forward(model, crit, batch)
    loss.backward()
    _, g64, _ = _oracle_grads(name, torch.float64)
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gsum = {k: v for k, v in zip(gold['grad_keys'].tolist(), gold['grad_sums'])}
    assert set(gsum) == {k for k, p in named.items() if p.grad is not None}
    scale = max(abs(v[1]) for v in gsum.values())
    e_hip, e_ref, e_hip_ref = [], [], []
    for k, ref in gsum.items():
        exact = float(g64[k].abs().sum())
        if exact < 1e-5 * scale:            # analytically-zero gradients (conv bias in front of a BN): noise
            continue
        mine = float(named[k].grad.detach().double().abs().sum())
        e_hip.append(abs(mine - exact) / exact)
        e_ref.append(abs(float(ref[1]) - exact) / exact)
        e_hip_ref.append(abs(mine - float(ref[1])) / abs(float(ref[1])))
    e_hip, e_ref, e_hip_ref = np.asarray(e_hip), np.asarray(e_ref), np.asarray(e_hip_ref)
    print("abs-sum of each parameter gradient, relative error: HIP vs fp64 median %.2e max %.2e | reference "
          "golden vs fp64 median %.2e max %.2e | HIP vs reference golden median %.2e max %.2e"
          % (np.median(e_hip), e_hip.max(), np.median(e_ref), e_ref.max(), np.median(e_hip_ref), e_hip_ref.max()))
    assert np.median(e_hip) <= max(1e-4, 3.0 * np.median(e_ref)), (np.median(e_hip), np.median(e_ref))
    assert e_hip.max() <= max(1e-3, 3.0 * e_ref.max()), (e_hip.max(), e_ref.max())
    # HIP vs the reference's numbers directly: two fp32 roundings of the same function differ by up to the
    # sum of their distances from it, parameter by parameter
    sl = 3.0 * (e_hip + e_ref) + 1e-4
    assert np.all(e_hip_ref <= sl), float((e_hip_ref / sl).max())


def test_headline_shape_eval_forward_vs_oracle(dev):
    """BASELINE configs[1] at its real geometry -- 64x2048x5, bi-LSTM 128x2, odometry bi-LSTM 1024x2,
    soft fusion -- B=1, S=2, eval mode: the shapes that switch the split-bf16 tile heuristics
    (conv_bx3.hip) and the 1x1 routing thresholds (functional._use_bx3), against the CPU oracle with the
    same fill_state weights: <= 1e-4 of the output scale (north_star)."""
    from deeplio_amd import misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    cfg = make_config(seq=2)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=1))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.eval()
    omodel = om.get_model((5, 64, 2048), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.eval()
    batch = gc.make_batch(7, 1, 2, 5, 64, 2048, 50)
    with torch.no_grad():
        pos, ori = model([[batch[0].to(dev), batch[1].to(dev)], batch[2].to(dev)])
        opos, oori = omodel([[batch[0], batch[1]], batch[2]])
    assert rel_err(pos, opos) < TOL and rel_err(ori, oori) < TOL, (rel_err(pos, opos), rel_err(ori, oori))


@pytest.mark.parametrize("B", [1, 8])
def test_headline_shape_train_forward_vs_oracle(dev, B):
    """same geometry, train mode (batch statistics over B*S images of 64x2048), dropout off: forward outputs, loss AND the
    running statistics of every BatchNorm against the CPU oracle <= 1e-4.  B = 8 is the launch size bench.py times
    (N = 16 images per encoder): there the one-launch BatchNorm of the small maps runs 16 waves, the cooperative kernels cut
    planes into parts, the Fire block in front of SELayer + pool pools its own output and the K-split heuristics take their
    other branches (trainer.py:238-266 of the reference is what is matched)"""
    from deeplio_amd import losses, misc, nets, ops
    from deeplio_amd.config import make_config
    from oracle import model as om
    from oracle import se3 as ose3
    cfg = make_config(seq=2, overrides=gc.NO_DROP)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    omodel = om.get_model((5, 64, 2048), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.train()
    ocrit = om.get_loss_function(cfg)
    batch = gc.make_batch(7, B, 2, 5, 64, 2048, 50)
    pt, pw, pp, pq, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    with torch.no_grad():
        a, b = omodel([[batch[0], batch[1]], batch[2]])
        p2, q2 = ose3.se3_to_SE3(a, b)
        oloss = ocrit(a, b, p2[:, 1:3], q2[:, 1:3], batch[3][:, :, 0:3], batch[3][:, :, 3:], batch[4][:, 1:3, 0:3],
                      batch[4][:, 1:3, 3:7])
    assert rel_err(pt, a) < TOL and rel_err(pw, b) < TOL, (rel_err(pt, a), rel_err(pw, b))
    assert rel_err(loss, oloss) < TOL
    # running statistics: every BatchNorm of both encoders saw the same batch statistics
    osd, worst = omodel.state_dict(), (0.0, None)
    n = 0
    for k, v in model.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            e = rel_err(v, osd[k])
            worst = max(worst, (e, k))
            n += 1
    assert n >= 2 * 2 * (1 + 3 * 12) and worst[0] < TOL, worst
    assert ops.bn_coop_errors() == 0


def test_lidar_fusion_cat_resnet_vs_oracle(dev):
    """BASELINE configs[3] as stated: lidar-feat-resnet with `fusion: cat` (+ bi-LSTM IMU net).  The
    reference crashes at construction for cat (fc1 hard-coded to the add width), so the
    behaviour is build-defined: fc1 = Linear(2F, 128) over [f_xyz || f_normals] -- the HIP path against the
    oracle's same definition: eval and train forward <= 1e-4, gradients inside the fp32 envelope."""
    from deeplio_amd import losses, misc, nets
    from oracle import model as om
    from oracle import se3 as ose3
    name = "resnet_lstm_cat"
    g = gc.MODEL_CASES[name]['geom']
    kw = dict(gc.MODEL_CASES[name]['cfg'])
    kw['overrides'] = dict(kw['overrides'], **{'lidar-feat-resnet/fusion': 'cat'})
    from deeplio_amd.config import make_config
    cfg = make_config(**kw)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=g['B']))
    model = nets.get_model((g['C'], g['H'], g['W']), cfg, dev)
    omodel = om.get_model((g['C'], g['H'], g['W']), cfg)
    assert model.lidar_feat_net.fc1.weight.shape == (128, 1024) == omodel.lidar_feat_net.fc1.weight.shape
    gc.fill_state(model, seed=1000)
    gc.fill_state(omodel, seed=1000)
    batch = gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
    dbatch = tuple(t.to(dev) for t in batch)
    model.eval(), omodel.eval()
    with torch.no_grad():
        pos, ori = model([[dbatch[0], dbatch[1]], dbatch[2]])
        opos, oori = omodel([[batch[0], batch[1]], batch[2]])
    assert rel_err(pos, opos) < TOL and rel_err(ori, oori) < TOL
    model.train(), omodel.train()
    crit, ocrit = losses.get_loss_function(cfg, dev), om.get_loss_function(cfg)
    pt, pw, pp, pq, loss = hip_step_forward(model, crit, dbatch)
    loss.backward()

    def ostep(dtype):
        m = om.get_model((g['C'], g['H'], g['W']), cfg)
        gc.fill_state(m, seed=1000)
        m, c = m.to(dtype).train(), om.get_loss_function(cfg).to(dtype)
        xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
        a, b = m([[xyz, nrm], imu])
        p2, q2 = ose3.se3_to_SE3(a, b)
        lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
        lo.backward()
        return a, lo, {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
    a32, l32, g32 = ostep(torch.float32)
    _, _, g64 = ostep(torch.float64)
    assert rel_err(pt, a32) < TOL and rel_err(loss, l32) < TOL
    named = dict(model.named_parameters())
    gmax = max(float(v.abs().max()) for v in g64.values())
    e_hip, e_ref = [], []
    for k, ref in g64.items():
        if float(ref.abs().max()) < 1e-5 * gmax:
            continue
        e_hip.append(_l2(named[k].grad.detach().double().cpu(), ref))
        e_ref.append(_l2(g32[k], ref))
    assert "lidar_feat_net.fc1.weight" in g64
    assert np.median(e_hip) <= max(1e-3, 3.0 * np.median(e_ref)), (np.median(e_hip), np.median(e_ref))
    assert max(e_hip) <= max(2e-2, 10.0 * max(e_ref)), (max(e_hip), max(e_ref))


def _headline_oracle_grads(cfg, batch, dtype):
    from oracle import model as om
    from oracle import se3 as ose3
    m = om.get_model((5, 64, 2048), cfg)
    gc.fill_state(m, seed=1000)
    m, c = m.to(dtype).train(), om.get_loss_function(cfg).to(dtype)
    xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
    a, b = m([[xyz, nrm], imu])
    p2, q2 = ose3.se3_to_SE3(a, b)
    lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    lo.backward()
    g = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
    g["criterion.sx"], g["criterion.sq"] = c.sx.grad.double(), c.sq.grad.double()
    return lo.detach().double(), g


def test_headline_shape_gradients_vs_oracle(dev):
    """BASELINE configs[1] at its real geometry (64x2048x5, bi-LSTM 128x2, odometry bi-LSTM 1024x2, soft fusion),
    B=1, S=2, train mode, dropout 0, fill_state weights: `loss.backward()` of the HIP path (trainer.py:238-266
    of the reference) against the CPU oracle in fp64, with the oracle's own fp32 run beside it.

    Measured (tools/grad_noise_probe.py, CPU oracle only): the envelope does NOT collapse at this size.  The
    gradient that reaches the output of the last Fire block is accurate in fp32 (1e-6 against fp64); one block
    further back torch's own fp32 gradient is 4e-3 (B = 8) ... 1.3e-2 (B = 1) away from fp64, and every encoder
    parameter inherits that.  Cause: ReLU / max-pool decisions.  The forward activations of the deepest blocks
    carry ~1e-5 relative fp32 error, so a fraction f ~ 1e-5 of the masks differs between ANY two fp32
    evaluations (or fp32 and fp64), and a masked gradient with a fraction f of its elements flipped is
    sqrt(f) ~ 3e-3 away in relative L2 -- per block, adding in quadrature on the way back (an upstream gradient
    without the average pool in front gives the same 1.3e-2).  No fp32 implementation can agree with another to
    1e-4 here, the reference with itself at another thread count included.  So: everything behind the encoders
    (IMU net, fusion, odometry net, heads, loss weights) is held to 1e-4; the encoder gradients must be samples
    of the same error distribution as torch-fp32's (median and maximum within 1.5x, every parameter within 5x
    of torch's own error for it); the 1e-4 claim for the encoder KERNELS at this geometry is carried per
    launch, on identical operands, by test_gpu_ops.py::test_conv_headline_launch_sizes_vs_fp64 and the
    decision-stable layer tests of test_gpu_modules.py."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    cfg = make_config(seq=2, overrides=gc.NO_DROP)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=1))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    batch = gc.make_batch(7, 1, 2, 5, 64, 2048, 50)
    *_, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    loss.backward()
    torch.cuda.synchronize()
    l64, g64 = _headline_oracle_grads(cfg, batch, torch.float64)
    l32, g32 = _headline_oracle_grads(cfg, batch, torch.float32)
    assert abs(float(loss.detach()) - float(l64)) <= TOL * abs(float(l64))
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, ref in g64.items():
        assert named[k].grad is not None, k
        mine = named[k].grad.detach().double().cpu()
        if float(ref.abs().max()) < 1e-5 * gmax:       # analytically-zero gradients (conv bias in front of a BN, Q3 weights)
            assert float(mine.abs().max()) < 1e-4 * gmax, k
            continue
        rows.append((k, _l2(mine, ref), _l2(g32[k], ref)))
    e_hip = np.asarray([r[1] for r in rows])
    e_ref = np.asarray([r[2] for r in rows])
    worst = sorted(rows, key=lambda r: -r[1] / max(r[2], 1e-4 / 3))[:6]
    print("headline gradients, rel-L2 vs fp64 over %d parameters: hip median %.2e max %.2e | torch-fp32 median %.2e "
          "max %.2e" % (len(rows), np.median(e_hip), e_hip.max(), np.median(e_ref), e_ref.max()))
    for k, a, b in worst:
        print("   %-60s hip %.2e  torch-fp32 %.2e" % (k, a, b))
    # everything behind the encoders (IMU net, fusion, odometry net, heads, loss weights) is well conditioned: 1e-4
    tail = [(k, a, b) for k, a, b in rows if not k.startswith("lidar_feat_net.encoder")]
    assert tail and all(a <= max(TOL, 3.0 * b) for _, a, b in tail), [r for r in tail if r[1] > max(TOL, 3.0 * r[2])][:8]
    assert np.median(e_hip) <= max(TOL, 1.5 * np.median(e_ref)), (np.median(e_hip), np.median(e_ref))
    assert e_hip.max() <= max(TOL, 1.5 * e_ref.max()), (e_hip.max(), e_ref.max())
    bad = [(k, a, b) for k, a, b in rows if a > max(TOL, 5.0 * b)]
    assert not bad, bad[:8]
    for k, p in model.named_parameters():
        if k not in g64:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("loss_type", ["local+global", "local", "global"])
def test_reference_iteration_protocol_on_hip_objects(dev, tmp_path, loss_type):
    """The call sequence of the reference's training iteration and epoch end (trainer.py:197-317 and :150-170),
    restated here, issued against deeplio_amd objects on the GPU and against the oracle's on the CPU: host-side
    NaN / Inf guards on the batch and the predictions, the detach-by-loss-type rule (:246-256), the criterion on the
    global window [1 : max_glob_seq + 1], `loss.detach().item()`, zero_grad / backward / step, calc_grad_norm over
    `model.parameters()` (:481-486), criterion.sx / sq reads, then `state_dict()` of model, optimizer and criterion
    into a checkpoint and PolynomialLRDecay.step().  Three iterations: losses and gradient norms agree with the
    oracle (1e-4 first iteration, the fp32 envelope afterwards), the checkpoint has the reference's layout and
    torch.optim loads its optimizer entry."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.optimizer import create_optimizer
    from deeplio_amd.se3 import se3_to_SE3
    from oracle import model as om
    from oracle import se3 as ose3
    name = "pointseg_lstm_cat"
    g = gc.MODEL_CASES[name]['geom']
    kw = dict(gc.MODEL_CASES[name]['cfg'])
    kw['overrides'] = dict(kw['overrides'], **{'losses/loss-type': loss_type})
    from deeplio_amd.config import make_config
    cfg = make_config(**kw)
    args = types.SimpleNamespace(device=str(dev), batch_size=g['B'], lr=1e-3, weight_decay=1e-4, momentum=0.9)
    misc.build_config_container(cfg, args)
    max_glob_seq = 2                                                       # trainer.py:42

    def calc_grad_norm(parameters):                                        # trainer.py:481-486
        ps = [p for p in parameters if p.grad is not None]
        return torch.norm(torch.stack([torch.norm(p.grad.detach(), 2) for p in ps]), 2)

    def guard(t, what):                                                    # trainer.py:221-243
        if torch.isnan(t).any() or torch.isinf(t).any():
            raise ValueError("%s:\n%s" % (what, t))

    def iteration(model, crit, opt, chain, batch):
        imgs, normals, imus, gts_f2f, gts_f2g = batch
        for t, what in ((gts_f2f, "gt-f2f"), (gts_f2g, "gt-f2g"), (normals, "normals"), (imgs, "imgs")):
            guard(t, what)
        gt_f2f_t, gt_f2f_w, gt_f2g_p, gt_f2g_q = gts_f2f[:, :, 0:3], gts_f2f[:, :, 3:], gts_f2g[:, :, 0:3], gts_f2g[:, :, 3:7]
        pred_f2f_t, pred_f2f_w = model([[imgs, normals], imus])
        guard(pred_f2f_t, "pred_f2f_x")
        guard(pred_f2f_w, "pred_f2f_r")
        pred_f2g_p, pred_f2g_q = chain(pred_f2f_t, pred_f2f_w)
        if crit.loss_Types[0] and not crit.loss_Types[1]:
            pred_f2g_p, pred_f2g_q = pred_f2g_p.detach(), pred_f2g_q.detach()
        elif not crit.loss_Types[0] and crit.loss_Types[1]:
            pred_f2f_t, pred_f2f_w = pred_f2f_t.detach(), pred_f2f_w.detach()
        sl = slice(1, max_glob_seq + 1)
        loss = crit(pred_f2f_t, pred_f2f_w, pred_f2g_p[:, sl, :], pred_f2g_q[:, sl, :], gt_f2f_t, gt_f2f_w,
                    gt_f2g_p[:, sl, :], gt_f2g_q[:, sl, :])
        val = loss.detach().item()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return val, float(calc_grad_norm(model.parameters())), float(crit.sx.data), float(crit.sq.data)

    model = nets.get_model((g['C'], g['H'], g['W']), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    opt = create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg, args)
    sched = misc.PolynomialLRDecay(opt, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    omodel = om.get_model((g['C'], g['H'], g['W']), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.train()
    ocrit = om.get_loss_function(cfg)
    oopt = torch.optim.Adam([{'params': omodel.parameters()}, {'params': ocrit.parameters()}], lr=1e-3, weight_decay=1e-4)
    from deeplio_amd import functional as Fh
    h2 = any(sw[0] for sw in (Fh._FIRE_H2, Fh._DGRAD_H2, Fh._DGRAD1_H2, Fh._WGRAD_H2, Fh._SMALL_H2))
    # new batch per iteration: the chaotic envelope of test_adam_trajectory, a little wider; the wider iteration-1 bound only
    # with the two-piece fp16 kernels on (fp32-level in norm, not per element: the sign-like first Adam step amplifies it)
    bounds = [1e-4, 4e-3 if h2 else 2e-3, 5e-2]
    # (iteration 1 sits at 1e-3 ... 2.5e-3 depending on which fp32-accurate kernel the tiny layers run on: the first Adam
    #  step is sign-like, see test_adam_trajectory)
    for it in range(3):
        batch = gc.make_batch(2000 + it, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])
        mine = iteration(model, crit, opt, se3_to_SE3, tuple(t.to(dev) for t in batch))
        ref = iteration(omodel, ocrit, oopt, ose3.se3_to_SE3, batch)
        assert abs(mine[0] - ref[0]) <= bounds[it] * abs(ref[0]), (it, mine, ref)
        # gradient norm: inside the fp32 gradient envelope of this tiny geometry (test_train_forward_backward_vs_oracle);
        # after an Adam step (sign-like first updates, test_adam_trajectory) the two fp32 trajectories are 10 % apart in it
        assert abs(mine[1] - ref[1]) <= (2e-2 if it == 0 else 0.2) * abs(ref[1]), (it, mine, ref)
        assert abs(mine[2] - ref[2]) <= 1e-5 + bounds[it] and abs(mine[3] - ref[3]) <= 1e-5 + bounds[it]
    # a bad batch is refused by the guards exactly as the reference does
    bad = [t.to(dev) for t in gc.make_batch(1, g['B'], g['S'], g['C'], g['H'], g['W'], g['T'])]
    bad[3][0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="gt-f2f"):
        iteration(model, crit, opt, se3_to_SE3, tuple(bad))
    # epoch end (trainer.py:150-170)
    state = {'epoch': 0, 'state_dict': model.state_dict(), 'best_acc': 1.0, 'optimizer': opt.state_dict(),
             'criterion': crit.state_dict()}
    path = str(tmp_path / "cpkt_deeplio.tar")
    torch.save(state, path)
    for feat_net in model.get_feat_networks():
        torch.save({'state_dict': feat_net.state_dict()}, str(tmp_path / ("cpkt_%s.tar" % feat_net.name)))
    sched.step()
    assert opt.param_groups[0]['lr'] < 1e-3
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {'epoch', 'state_dict', 'best_acc', 'optimizer', 'criterion'}
    assert set(ck['state_dict']) == set(omodel.state_dict()) and set(ck['criterion']) == set(ocrit.state_dict())
    oopt.load_state_dict(ck['optimizer'])                                   # the reference resumes from it
    omodel.load_state_dict(ck['state_dict'])


@pytest.mark.parametrize("lidar,imu_type,C", [("lidar-feat-flownet", "gru", 3), ("lidar-feat-resnet", "lstm", 3)])
def test_full_size_eval_forward_other_families_vs_oracle(dev, lidar, imu_type, C):
    """BASELINE configs[2] / configs[3] at their real geometry (64x2048, C = 3 per stream: xyz + normals as kitti.py
    delivers them), B = 1, S = 2, eval mode, fill_state weights: FlowNet (lidar_feat_nets.py:240-267: 5x7 / 3x5 strided
    stems, 3x3 stride-2 layers, 1024 channels at 8x32) and ResNet (resnet.py:14-112: 5x7 stem at full resolution,
    BasicBlocks with (1,2) / (2,2) downsampling) against the CPU oracle <= 1e-4 of the output scale -- the launch sizes
    at which the strided kernels, the phase-decomposed data gradients' forward twins and the 512 / 1024-channel 3x3 layers
    pick their large-tile branches."""
    from deeplio_amd import misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    cfg = make_config(lidar=lidar, imu="imu-feat-rnn", fusion="fusion-layer-cat", odom="odom-feat-rnn", seq=2,
                      overrides=dict(gc.NO_DROP, **{'imu-feat-rnn/type': imu_type}))
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=1))
    model = nets.get_model((C, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.eval()
    omodel = om.get_model((C, 64, 2048), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.eval()
    batch = gc.make_batch(9, 1, 2, C, 64, 2048, 50)
    with torch.no_grad():
        pos, ori = model([[batch[0].to(dev), batch[1].to(dev)], batch[2].to(dev)])
        opos, oori = omodel([[batch[0], batch[1]], batch[2]])
    assert rel_err(pos, opos) < TOL and rel_err(ori, oori) < TOL, (rel_err(pos, opos), rel_err(ori, oori))


@pytest.mark.parametrize("lidar,imu_type,C", [("lidar-feat-flownet", "gru", 3), ("lidar-feat-resnet", "lstm", 3)])
def test_full_size_train_step_other_families_vs_oracle(dev, lidar, imu_type, C):
    """BASELINE configs[2] / configs[3] at their real geometry (64x2048, C = 3 per stream), B = 2, S = 2, TRAIN mode: one
    training step (train-mode BatchNorm at the real launch sizes -- the cooperative one-launch kernels where the planes are
    large -- strided split-bf16 forwards, dropout off), predictions, SE(3) chain, HWS loss and the updated running
    statistics against the CPU oracle <= 1e-4; then loss.backward() through the phase-decomposed data gradients and the
    tap-wise stride-2 weight gradients: every parameter that the oracle gives a gradient gets a finite one, and everything
    behind the encoders (IMU net, fusion, odometry net, heads, loss weights -- no ReLU / max-pool decisions of the deep
    convolution stack in front of them) agrees with the oracle's fp32 gradients to 1e-4 of the gradient scale
    (lidar_feat_nets.py:119-148,165-189,240-267; resnet.py:14-112; trainer.py:263-281)."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    from oracle import se3 as ose3
    cfg = make_config(lidar=lidar, imu="imu-feat-rnn", fusion="fusion-layer-cat", odom="odom-feat-rnn", seq=2,
                      overrides=dict(gc.NO_DROP, **{'imu-feat-rnn/type': imu_type}))
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=2))
    model = nets.get_model((C, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    omodel = om.get_model((C, 64, 2048), cfg)
    gc.fill_state(omodel, seed=1000)
    omodel.train()
    ocrit = om.get_loss_function(cfg)
    batch = gc.make_batch(11, 2, 2, C, 64, 2048, 50)
    pt, pw, pp, pq, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    loss.backward()
    xyz, nrm, imu, gt_f2f, gt_f2g = batch
    opt_, opw = omodel([[xyz, nrm], imu])
    opp, opq = ose3.se3_to_SE3(opt_, opw)
    oloss = ocrit(opt_, opw, opp[:, 1:3], opq[:, 1:3], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:], gt_f2g[:, 1:3, 0:3], gt_f2g[:, 1:3, 3:7])
    oloss.backward()
    assert rel_err(pt, opt_) < TOL and rel_err(pw, opw) < TOL and rel_err(pp, opp) < TOL and rel_err(pq, opq) < TOL
    assert rel_err(loss, oloss) < TOL, (float(loss), float(oloss))
    obufs = dict(omodel.named_buffers())
    for k, b in model.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(b, obufs[k]) < 1e-5, k
    ograds = {k: p.grad for k, p in omodel.named_parameters() if p.grad is not None}
    gmax = max(float(v.abs().max()) for v in ograds.values())
    named = dict(model.named_parameters())
    behind = [k for k in ograds if not k.startswith("lidar_feat_net.")]
    assert behind and len(behind) < len(ograds)
    for k, ref in ograds.items():
        g = named[k].grad
        assert g is not None and bool(torch.isfinite(g).all()), k
        if k in behind:
            assert float((g.cpu() - ref).abs().max()) <= 1e-4 * gmax, (k, float((g.cpu() - ref).abs().max()) / gmax)


# ---- decision-pinned gradient check (test infrastructure only: nothing of this lives in deeplio_amd/) ----------------------
def _graph_nodes(root):
    seen, stack, out = set(), [root], []
    while stack:
        n = stack.pop()
        if n is None or n in seen:
            continue
        seen.add(n)
        out.append(n)
        stack.extend(f for f, _ in n.next_functions)
    return out


def _relu_mask(raw, prm, beta):
    """the ReLU decisions of a BatchNorm + ReLU layer as the library's BACKWARD takes them: eval-mode BatchNorm backward of
    dy = 1 is dx = scale * [activated output > 0] (bn.hip: the mask is re-derived from the raw convolution output)"""
    from deeplio_amd import ops
    raw = raw.contiguous()
    N, C_, H, W = raw.shape
    dx = torch.empty_like(raw)
    p = tuple(t.contiguous() for t in prm)
    ops.bn_bwd_fused(torch.ones_like(raw), C_, 0, raw, C_, 0, p, beta, dx, C_, 0, N, C_, H * W, False, True, False)
    assert float(p[2].abs().min()) > 0
    return (dx != 0).cpu()


def _hip_decisions(model, loss):
    """{(encoder, block, index) -> masks / arg-max maps} read off the tape of a HIP forward: FireFn keeps the raw squeeze and
    expand outputs and their BatchNorm parameters, the stem node and the SELayer + pool nodes keep their uint8 arg-max maps"""
    ptr = {p.data_ptr(): k for k, p in model.named_parameters()}
    dec = {}
    for node in _graph_nodes(loss.grad_fn):
        name = type(node).__name__
        if name == "FireFnBackward":
            t = node.saved_tensors
            key = tuple(ptr[t[1].data_ptr()].split(".")[1:4])                 # (encoderN, fire_blkM, i)
            raw_s, raw_e, prm_s = t[7], t[9], t[10]
            E1 = t[3].shape[0]
            if node.cfg[5]:                                                    # apply-on-load block: (aff, inv1, inv3)
                aff, inv1, inv3 = t[11], t[12], t[13]
                prm_1, prm_3 = (aff[0, :E1], inv1, aff[1, :E1]), (aff[0, E1:], inv3, aff[1, E1:])
            else:
                prm_1, prm_3 = tuple(t[11]), tuple(t[12])
            dec[key] = dict(s=_relu_mask(raw_s, tuple(prm_s), t[2]), e1=_relu_mask(raw_e[:, :E1], prm_1, t[4]),
                            e3=_relu_mask(raw_e[:, E1:], prm_3, t[6]))
        elif name == "ConvBnActPoolFnBackward":
            t = node.saved_tensors                                             # x, weight, beta, raw, aff, inv, gamma, bias, idx
            key = (ptr[t[1].data_ptr()].split(".")[1], "stem")
            dec[key] = dict(mask=_relu_mask(t[3], (t[4][0], t[5], t[4][1]), t[2]), idx=t[8].cpu())
        elif name == "SEPoolFnBackward" and node.pool is not None:
            t = node.saved_tensors                                             # x, w1, w2, g, h, s, idx
            key = tuple(ptr[t[1].data_ptr()].split(".")[1:3]) + ("pool",)
            dec[key] = dict(idx=t[6].cpu())
        elif name == "ConvBnActBackward":                                      # plain conv + BN (+ ReLU) layers: FlowNet, ResNet
            t = node.saved_tensors                                             # x, weight, beta, raw, prm, gamma, bias
            if node.cfg[3]:                                                    # post_relu
                dec[("cbr", ptr[t[1].data_ptr()][:-len(".weight")])] = dict(mask=_relu_mask(t[3], tuple(t[4]), t[2]))
        elif name == "MaxPoolFnBackward":                                      # (ResNet: one per encoder, behind conv1 + bn1)
            src = node.next_functions[0][0]
            key = ("pool", ptr[src.saved_tensors[1].data_ptr()].split(".")[1])
            dec[key] = dict(idx=node.saved_tensors[0].cpu())
        elif name == "BinaryFnBackward" and node.op == 3:                      # BasicBlock tail relu(bn2(conv2(.)) + identity)
            src = node.next_functions[0][0]                                    # the bn2 node
            blk = ptr[src.saved_tensors[1].data_ptr()][:-len(".conv2.weight")]
            dec[("tail", blk)] = dict(mask=(node.saved_tensors[0] > 0).cpu())
    return dec


def _pool_by_index(x, idx, stride):
    """3x3 / pad 1 max-pool with the window element given (uint8 ky * 3 + kx, pool.hip) instead of decided"""
    N, C_, H, W = x.shape
    OH, OW = idx.shape[2], idx.shape[3]
    idx = idx.long()
    ih = torch.arange(OH).view(1, 1, OH, 1) * stride[0] - 1 + idx // 3
    iw = torch.arange(OW).view(1, 1, 1, OW) * stride[1] - 1 + idx % 3
    assert int(ih.min()) >= 0 and int(ih.max()) < H and int(iw.min()) >= 0 and int(iw.max()) < W
    return x.flatten(2).gather(2, (ih * W + iw).flatten(2)).view(N, C_, OH, OW)


def _pinned_encoder_forward(enc, ename, dec):
    """PSEncoder.forward of the oracle (pointseg_net.py:57-71, pointseg_modules.py:116-142) with every ReLU replaced by the
    given mask and every max-pool by the given arg-max"""
    from oracle import model as om

    def fwd(x):
        st = dec[(ename, "stem")]
        x = enc.conv1a[1](enc.conv1a[0](x)) * st["mask"].to(x.dtype)
        x = _pool_by_index(x, st["idx"], (1, 2))
        for name, fires, se, pool in om.PS_BLOCKS:
            i = 0
            for m in getattr(enc, name):
                if isinstance(m, om.Fire):
                    d = dec[(ename, name, str(i))]
                    s = m.squeeze_bn(m.squeeze(x)) * d["s"].to(x.dtype)
                    a = m.expand1x1_bn(m.expand1x1(s)) * d["e1"].to(x.dtype)
                    b = m.expand3x3_bn(m.expand3x3(s)) * d["e3"].to(x.dtype)
                    out = torch.cat([a, b], 1)
                    x = out + x if m.residual else out
                    i += 1
                elif isinstance(m, om.SELayer):
                    x = m(x)
                else:
                    x = _pool_by_index(x, dec[(ename, name, "pool")]["idx"], pool)
        return x
    return fwd


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


@pytest.mark.parametrize("B", [1, 4, 8])
def test_headline_encoder_gradients_with_the_decisions_pinned(dev, B):
    """The envelope of test_headline_shape_gradients_vs_oracle turned into a test of the kernels: at the headline geometry
    (64x2048x5, B = 1, B = 4 -- N = 8 images per encoder: planes cut into parts -- and B = 8: N = 16, THE LAUNCH bench.py times
    (part counts of the cooperative BatchNorm kernels, K-splits and slab counts of the weight gradients are those of the timed
    step), S = 2, train mode) the fp64 oracle (at B = 8 on a host without ~150 GB of free memory: the fp32 oracle with the same
    pinned decisions, itself measured 7e-5 from fp64 -- the bound is then 2e-4 = both sides' distance from fp64) is run with the ReLU masks and max-pool arg-max maps THE HIP
    FORWARD USED (read off its tape: raw convolution outputs + BatchNorm parameters through the library's own backward
    mask, the uint8 arg-max maps) instead of deciding them itself.  With the decisions equal, the 1e-2 disagreement of any
    two fp32 evaluations of this network is gone and `loss.backward()` of the HIP path (trainer.py:263-281) must match the
    oracle per parameter: <= 1e-4 relative L2 for every parameter, encoders included -- except a gradient that is a heavily
    cancelling sum (the stem's BatchNorm bias at B = 4: 1.8e-4 here, 7e-5 for torch fp32), which is held to three times the
    error of the reference's own fp32 arithmetic under the same pinned decisions (measured: switching any of the two-piece
    kernels back to three pieces leaves it between 1.3e-4 and 1.8e-4)."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    from oracle import se3 as ose3
    cfg = make_config(seq=2, overrides=gc.NO_DROP)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    batch = gc.make_batch(7, B, 2, 5, 64, 2048, 50)
    *_, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    dec = _hip_decisions(model, loss)
    assert len(dec) == 2 * (12 + 1 + 4), sorted(dec)            # per encoder: 12 Fire blocks, the stem, 4 SELayer + pool
    loss.backward()
    torch.cuda.synchronize()
    def pinned_oracle(dtype):
        m = om.get_model((5, 64, 2048), cfg)
        gc.fill_state(m, seed=1000)
        m, c = m.to(dtype).train(), om.get_loss_function(cfg).to(dtype)
        for ename in ("encoder1", "encoder2"):
            enc = getattr(m.lidar_feat_net, ename)
            enc.forward = _pinned_encoder_forward(enc, ename, dec)
        xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
        a, b = m([[xyz, nrm], imu])
        p2, q2 = ose3.se3_to_SE3(a, b)
        lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
        lo.backward()
        g = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
        g["criterion.sx"], g["criterion.sq"] = c.sx.grad.double(), c.sq.grad.double()
        return float(lo.detach()), g
    # fp64 activations of the pinned oracle: ~4.5 GB per frame pair per encoder pair (B = 4: ~40 GB; B = 8: ~75 GB + the masks)
    ref64 = B < 8 or _mem_available_gb() > 150.0
    if B == 8 and not ref64 and _mem_available_gb() < 60.0:
        pytest.skip("the pinned oracle at B = 8 needs ~40 GB of host memory in fp32 (%.0f GB available)" % _mem_available_gb())
    tol = TOL if ref64 else 2.0 * TOL
    lo, g64 = pinned_oracle(torch.float64 if ref64 else torch.float32)
    print("pinned oracle: %s (MemAvailable %.0f GB)" % ("fp64" if ref64 else "fp32", _mem_available_gb()))
    assert abs(float(loss.detach()) - lo) <= TOL * abs(lo)
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, ref in g64.items():
        mine = named[k].grad.detach().double().cpu()
        if float(ref.abs().max()) < 1e-5 * gmax:
            assert float(mine.abs().max()) < 1e-4 * gmax, k
            continue
        rows.append((k, _l2(mine, ref)))
    errs = np.asarray([r[1] for r in rows])
    enc_rows = [r for r in rows if r[0].startswith("lidar_feat_net.encoder")]
    print("pinned decisions, rel-L2 vs fp64 over %d parameters (%d in the encoders): median %.2e max %.2e"
          % (len(rows), len(enc_rows), np.median(errs), errs.max()))
    for k, e in sorted(rows, key=lambda r: -r[1])[:5]:
        print("   %-64s %.2e" % (k, e))
    assert len(enc_rows) > 200
    over = [r for r in rows if r[1] > tol]
    if over and not ref64:
        # (no fp64 yardstick on this host: the one parameter the fp64 branch waives -- the stem's BatchNorm bias, 1.5-2.2e-4 from
        #  fp64 at B = 8 -- is held to 3e-4 + the fp32 oracle's own 1e-4; anything else over the tolerance fails)
        bad = [r for r in over if not (r[0].endswith("conv1a.1.bias") and r[1] <= 4e-4)]
        if bad:
            raise AssertionError("over %.0e against the fp32 pinned oracle: %r" % (tol, bad))
        over = []
    if over:
        # A parameter gradient that is a heavily cancelling sum (the stem's BatchNorm bias at N = 8 images: the gradient field
        # behind a BatchNorm backward has zero mean per channel, d loss / d beta sums 4 M signed terms of it) is not computable
        # to 1e-4 in fp32 by ANY evaluation order.  For those -- and only those -- the yardstick is the reference's own
        # arithmetic: the oracle in fp32 with the same pinned decisions, against fp64.
        _, g32 = pinned_oracle(torch.float32)
        for k, e in over:
            e32 = _l2(g32[k], g64[k])
            print("   over 1e-4: %-52s HIP %.2e | torch fp32 with the same decisions %.2e" % (k, e, e32))
            # (measured: HIP 1.8e-4 at B = 4, 2.2e-4 at B = 8, the same on every box; the fp32 yardstick itself moves with the
            #  host's core count -- its reduction tree -- 5.0e-5 .. 7.5e-5 over the boxes this ran on, so the ratio was seen at
            #  2.0x .. 3.7x.  The split-operand kernels' errors are relative to the TENSOR's scale, torch's to each element, and
            #  this sum of 4 - 8 M signed terms cancels to ~1e-3 of their magnitude.  Held to 5x the yardstick AND to 3e-4.)
            assert e <= 5.0 * e32 and e <= 3e-4, (k, e, e32)
        assert len(over) <= 4 and all(k.endswith("conv1a.1.bias") for k, _ in over), over


def test_reference_arithmetic_disagrees_with_itself_across_thread_counts(dev):
    """What the envelope tests rest on, as a measurement instead of a sentence: the SAME torch fp32 code (the oracle =
    the reference's arithmetic) at the headline geometry with 1 and with 8 intra-op threads -- different reduction trees in
    the convolutions -- gives encoder gradients that differ from each other by far more than 1e-4 (ReLU / max-pool
    decision flips), while everything behind the encoders agrees.  No fp32 implementation can be held to 1e-4 against
    another on the encoder gradients without pinning the decisions
    (test_headline_encoder_gradients_with_the_decisions_pinned does that)."""
    from deeplio_amd.config import make_config
    cfg = make_config(seq=2, overrides=gc.NO_DROP)
    batch = gc.make_batch(7, 1, 2, 5, 64, 2048, 50)
    keep = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        _, g1 = _headline_oracle_grads(cfg, batch, torch.float32)
        torch.set_num_threads(8)
        _, g8 = _headline_oracle_grads(cfg, batch, torch.float32)
    finally:
        torch.set_num_threads(keep)
    gmax = max(float(v.abs().max()) for v in g8.values())
    enc, rest = [], []
    for k, ref in g8.items():
        if float(ref.abs().max()) < 1e-5 * gmax:
            continue
        (enc if k.startswith("lidar_feat_net.encoder") else rest).append(_l2(g1[k], ref))
    enc, rest = np.asarray(enc), np.asarray(rest)
    print("torch fp32, 1 thread vs 8 threads: encoder gradients rel-L2 median %.2e max %.2e | behind the encoders median %.2e "
          "max %.2e" % (np.median(enc), enc.max(), np.median(rest), rest.max()))
    assert np.median(enc) > 10 * TOL and rest.max() < 10 * TOL


def _pinned_flownet_forward(enc, prefix, dec):
    """FlowNetEncoder.forward of the oracle (lidar_feat_nets.py:240-267) with every ReLU replaced by the given mask"""
    from oracle import model as om

    def fwd(x):
        for name, *_ in om.FLOWNET_LAYERS:
            seq = getattr(enc, name)
            x = seq[1](seq[0](x)) * dec[("cbr", "%s.%s.0" % (prefix, name))]["mask"].to(x.dtype)
        return x.mean((2, 3))
    return fwd


def _pinned_resnet_forward(enc, prefix, dec):
    """ResNetEncoder.forward of the oracle (resnet.py:94-112 + torchvision BasicBlock) with the ReLU masks of conv1 / every
    block's first convolution / every block tail and the max-pool's arg-max map given instead of decided"""
    def fwd(x):
        x = enc.bn1(enc.conv1(x)) * dec[("cbr", prefix + ".conv1")]["mask"].to(x.dtype)
        x = _pool_by_index(x, dec[("pool", prefix.split(".")[1])]["idx"], (1, 2))
        for i in range(4):
            for j, blk in enumerate(getattr(enc, "layer%d" % (i + 1))):
                p = "%s.layer%d.%d" % (prefix, i + 1, j)
                idt = x if blk.downsample is None else blk.downsample(x)
                out = blk.bn1(blk.conv1(x)) * dec[("cbr", p + ".conv1")]["mask"].to(x.dtype)
                out = blk.bn2(blk.conv2(out))
                x = (out + idt) * dec[("tail", p)]["mask"].to(x.dtype)
        return x.mean((2, 3))
    return fwd


@pytest.mark.parametrize("lidar,imu_type,C", [("lidar-feat-flownet", "gru", 3), ("lidar-feat-resnet", "lstm", 3)])
def test_other_families_encoder_gradients_with_the_decisions_pinned(dev, lidar, imu_type, C):
    """BASELINE configs[2] / configs[3] at their real geometry (64x2048, C = 3 per stream), B = 2, S = 2 (N = 4 images per
    encoder), train mode: `loss.backward()` of the HIP path (trainer.py:263-281) against the fp64 oracle run with the ReLU
    masks (conv + BN + ReLU layers, BasicBlock tails) and the max-pool arg-max map THE HIP FORWARD USED -- every parameter,
    the encoders included (lidar_feat_nets.py:240-267: FlowNet's 5x7 / 3x5 strided stems, 3x3 stride-2 layers; resnet.py:14-112:
    the 5x7 full-resolution stem, BasicBlocks with (1, 2) / (2, 2) downsampling): the phase-decomposed data gradients, the
    tap-wise stride-2 weight gradients and the two-piece fp16 layers are held to <= 1e-4 relative L2 per parameter (a heavily
    cancelling sum -- a BatchNorm bias behind millions of signed terms -- to three times the error of torch fp32 under the
    same pinned decisions, as in the PointSeg test above)."""
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    from oracle import se3 as ose3
    B = 2
    cfg = make_config(lidar=lidar, imu="imu-feat-rnn", fusion="fusion-layer-cat", odom="odom-feat-rnn", seq=2,
                      overrides=dict(gc.NO_DROP, **{'imu-feat-rnn/type': imu_type}))
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    model = nets.get_model((C, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    batch = gc.make_batch(11, B, 2, C, 64, 2048, 50)
    *_, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    dec = _hip_decisions(model, loss)
    flownet = lidar == "lidar-feat-flownet"
    assert len(dec) == (2 * 9 if flownet else 2 * (1 + 1 + 11 + 11)), sorted(dec)     # ResNet: conv1, pool, 11 x (conv1, tail)
    loss.backward()
    torch.cuda.synchronize()

    def pinned_oracle(dtype):
        m = om.get_model((C, 64, 2048), cfg)
        gc.fill_state(m, seed=1000)
        m, c = m.to(dtype).train(), om.get_loss_function(cfg).to(dtype)
        for ename in ("encoder1", "encoder2"):
            enc = getattr(m.lidar_feat_net, ename)
            enc.forward = (_pinned_flownet_forward if flownet else _pinned_resnet_forward)(enc, "lidar_feat_net." + ename, dec)
        xyz, nrm, imu, f2f, f2g = (t.to(dtype) for t in batch)
        a, b = m([[xyz, nrm], imu])
        p2, q2 = ose3.se3_to_SE3(a, b)
        lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
        lo.backward()
        g = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
        g["criterion.sx"], g["criterion.sq"] = c.sx.grad.double(), c.sq.grad.double()
        return float(lo.detach()), g
    lo, g64 = pinned_oracle(torch.float64)
    assert abs(float(loss.detach()) - lo) <= TOL * abs(lo)
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, ref in g64.items():
        mine = named[k].grad.detach().double().cpu()
        if float(ref.abs().max()) < 1e-5 * gmax:
            assert float(mine.abs().max()) < 1e-4 * gmax, k
            continue
        rows.append((k, _l2(mine, ref)))
    errs = np.asarray([r[1] for r in rows])
    enc_rows = [r for r in rows if r[0].startswith("lidar_feat_net.encoder")]
    print("%s, pinned decisions, rel-L2 vs fp64 over %d parameters (%d in the encoders): median %.2e max %.2e"
          % (lidar, len(rows), len(enc_rows), np.median(errs), errs.max()))
    for k, e in sorted(rows, key=lambda r: -r[1])[:5]:
        print("   %-64s %.2e" % (k, e))
    assert len(enc_rows) >= (2 * 9 * 3 - 4 if flownet else 100)
    over = [r for r in rows if r[1] > TOL]
    if over:
        _, g32 = pinned_oracle(torch.float32)
        for k, e in over:
            e32 = _l2(g32[k], g64[k])
            print("   over 1e-4: %-52s HIP %.2e | torch fp32 with the same decisions %.2e" % (k, e, e32))
            assert e <= 5.0 * e32 and e <= 3e-4, (k, e, e32)      # (measured: ResNet's stem BatchNorm bias 2.6x, a sum of 0.5 M cancelling terms; the yardstick moves with the host core count)
        assert len(over) <= 6 and all(".bias" in k or "bn" in k for k, _ in over), over


def _imu_forward_with_masks(net, masks, p):
    """ImufeatRNN0.forward of the oracle (imu_feat_nets.py:75-83) with nn.LSTM's inter-layer dropout (drawn inside the fused
    nn.LSTM call, not injectable) replaced by the given masks: the two layers run as two single-layer bidirectional nn.LSTMs
    that SHARE the 2-layer module's Parameters (gradients land in the oracle model), state carried per layer from sub-sequence
    s to s + 1 exactly as the stacked call carries it"""
    rnn = net.rnn
    H = rnn.hidden_size
    dtype = rnn.weight_ih_l0.dtype

    def layer(l):
        m = torch.nn.LSTM(rnn.input_size if l == 0 else 2 * H, H, 1, bidirectional=True, batch_first=True).to(dtype)
        for sfx in ("", "_reverse"):
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                setattr(m, "%s_l0%s" % (nm, sfx), getattr(rnn, "%s_l%d%s" % (nm, l, sfx)))
        return m
    l0, l1 = layer(0), layer(1)

    def fwd(x):
        b, s, t, n = x.shape
        st0 = st1 = None
        outs = []
        for i in range(s):
            o0, st0 = l0(x[:, i], st0)
            o0 = o0 * masks[i].view(b, t, 2 * H).to(o0.dtype) / (1.0 - p)
            o1, st1 = l1(o0, st1)
            outs.append(o1.view(b, t, 2, H)[:, -1, 0, :])
        return torch.stack(outs, 1)
    return fwd


def test_headline_step_with_dropout_on_and_the_masks_copied_into_the_oracle(dev):
    """BASELINE configs[1] as bench.py times it -- dropout ON (lidar head 0.1, nn.LSTM inter-layer 0.1 in the IMU net, 0.25 in
    front of the heads: lidar_feat_nets.py:97-99, imu_feat_nets.py:62-66, deeplio_nets.py:84-86) -- at 64x2048x5, B = 1, S = 2,
    train mode: the masks the HIP kernels drew (Philox, deeplio_amd.functional._dropout_launch) are copied into the fp64 oracle
    together with the encoders' ReLU / arg-max decisions; loss and EVERY parameter gradient then agree to 1e-4 -- the scaling by
    1 / (1 - p), the mask re-use in backward and the RNN's inter-layer site included."""
    from deeplio_amd import functional as Fh
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from oracle import model as om
    from oracle import se3 as ose3
    B = 1
    cfg = make_config(seq=2)
    assert cfg['deeplio']['dropout'] == 0.25 and cfg['lidar-feat-pointseg']['dropout'] == 0.1 and cfg['imu-feat-rnn']['dropout'] == 0.1
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    model = nets.get_model((5, 64, 2048), cfg, dev)
    gc.fill_state(model, seed=1000)
    model.train()
    crit = losses.get_loss_function(cfg, dev)
    batch = gc.make_batch(7, B, 2, 5, 64, 2048, 50)
    drawn = []
    launch = Fh._dropout_launch

    def recording(x, p):
        y, mask = launch(x, p)
        drawn.append((tuple(x.shape), p, mask))
        return y, mask
    heads = Fh.ops.heads_fwd

    def heads_recording(x, ldx, R, K, wp, bp, wo, bo, p, seed, offset):
        # (the dropout in front of the heads is drawn inside functional.HeadsFn's launch, same Philox position)
        out = heads(x, ldx, R, K, wp, bp, wo, bo, p, seed, offset)
        drawn.append(((B, 2, K), p, out[2]))
        return out
    Fh._dropout_launch = recording
    Fh.ops.heads_fwd = heads_recording
    try:
        *_, loss = hip_step_forward(model, crit, tuple(t.to(dev) for t in batch))
    finally:
        Fh._dropout_launch = launch
        Fh.ops.heads_fwd = heads
    # issue order (nets.DeepLIO.forward_features / forward_tail): the IMU net's S inter-layer sites, the lidar head, the heads
    assert [(s, p) for s, p, _ in drawn] == [((B * 50, 256), 0.1)] * 2 + [((B * 2, 128), 0.1), ((B, 2, 1024), 0.25)], drawn
    masks = [m.cpu() for _, _, m in drawn]
    for m, (_, p, _) in zip(masks, drawn):
        assert set(m.unique().tolist()) <= {0, 1} and abs(float(m.float().mean()) - (1 - p)) < 0.1
    dec = _hip_decisions(model, loss)
    loss.backward()
    torch.cuda.synchronize()
    m = om.get_model((5, 64, 2048), cfg)
    gc.fill_state(m, seed=1000)
    m, c = m.double().train(), om.get_loss_function(cfg).double()
    for ename in ("encoder1", "encoder2"):
        enc = getattr(m.lidar_feat_net, ename)
        enc.forward = _pinned_encoder_forward(enc, ename, dec)
    m.imu_feat_net.forward = _imu_forward_with_masks(m.imu_feat_net, masks[0:2], 0.1)
    m.lidar_feat_net.drop.forward = lambda x: x * masks[2].view(x.shape).to(x.dtype) / 0.9
    m.drop.forward = lambda x: x * masks[3].view(x.shape).to(x.dtype) / 0.75
    xyz, nrm, imu, f2f, f2g = (t.double() for t in batch)
    a, b = m([[xyz, nrm], imu])
    p2, q2 = ose3.se3_to_SE3(a, b)
    lo = c(a, b, p2[:, 1:3], q2[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) <= TOL * abs(float(lo.detach()))
    g64 = {k: p.grad.double() for k, p in m.named_parameters() if p.grad is not None}
    g64["criterion.sx"], g64["criterion.sq"] = c.sx.grad.double(), c.sq.grad.double()
    named = dict(model.named_parameters())
    named["criterion.sx"], named["criterion.sq"] = crit.sx, crit.sq
    gmax = max(float(v.abs().max()) for v in g64.values())
    rows = []
    for k, ref in g64.items():
        mine = named[k].grad.detach().double().cpu()
        if float(ref.abs().max()) < 1e-5 * gmax:
            assert float(mine.abs().max()) < 1e-4 * gmax, k
            continue
        rows.append((k, _l2(mine, ref)))
    errs = np.asarray([r[1] for r in rows])
    print("dropout on, masks + decisions pinned: rel-L2 vs fp64 over %d parameters: median %.2e max %.2e (%s)"
          % (len(rows), np.median(errs), errs.max(), max(rows, key=lambda r: r[1])[0]))
    assert len(rows) > 250 and errs.max() <= TOL, sorted(rows, key=lambda r: -r[1])[:5]


def test_train_step_polls_its_error_words_without_being_asked(dev):
    """TrainStep.step copies the device-side error words (non-finite output, det != 1, a cooperative BatchNorm launch at its
    spin limit) into pinned memory every `check_every` steps and inspects the copy made one period earlier: a bad step raises
    within 2 x check_every steps although nobody calls check() (trainer.py:240-243 raises in the same iteration -- with a host
    sync per step)."""
    from deeplio_amd.trainer import TrainStep
    name = "pointseg_lstm_cat"
    g = gc.MODEL_CASES[name]['geom']
    ts = TrainStep(gc.case_cfg(name), (g['C'], g['H'], g['W']), dev, g['B'])
    ts.check_every = 2
    batch = tuple(t.to(dev) for t in gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T']))
    try:
        for _ in range(6):
            ts.step(*batch)                       # healthy steps: the polls find nothing
        bad = list(batch)
        bad[2] = bad[2].clone()                   # (the IMU stream: no ReLU between it and the heads -- fmaxf(NaN, 0) = 0 in the
        bad[2][0, 0, 0, 0] = float("nan")         #  encoders' BatchNorm + ReLU kernels swallows a NaN pixel)
        ts.step(*bad)                             # the model output of this step is non-finite
        with pytest.raises(ValueError, match="non-finite"):
            for _ in range(2 * ts.check_every + 1):
                ts.step(*batch)
    finally:
        ts.release_gc()


@pytest.mark.parametrize("optim", ["adam", "sgd"])
def test_early_tail_optimizer_step_is_bit_identical(dev, optim):
    """TrainStep issues the optimizer sweep over everything behind the feature nets (odometry net, heads, loss weights: their
    gradients are final when backward reaches the fusion output) from inside backward, on the 'comm' stream, and the rest at
    the end of the step (optimizer.FlatOptimizer.step_early): element-wise update rules, so parameters, moments and losses
    after three steps are BIT-identical to the single sweep at the end (create_optimizer, optimizer.py:4-16; trainer.py:263-266)."""
    from deeplio_amd.trainer import TrainStep
    name = "pointseg_lstm_cat"
    g = gc.MODEL_CASES[name]['geom']
    batch = tuple(t.to(dev) for t in gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T']))
    res = []
    for early in (True, False):
        cfg = gc.case_cfg(name)
        cfg['optimizer'] = optim
        from deeplio_amd import functional as Fh
        Fh.manual_seed(11)
        ts = TrainStep(cfg, (g['C'], g['H'], g['W']), dev, g['B'])
        gc.fill_state(ts.model, seed=1000)
        ts.early_tail_step = early
        assert ts._tail_lo is not None and 0 < ts._tail_lo < ts.optimizer.flat.numel()
        losses = [float(ts.step(*batch)) for _ in range(3)]
        ts.check()
        torch.cuda.synchronize()
        assert ts.optimizer.step_count == 3 and ts.optimizer._early is None
        st = ts.optimizer._state()
        res.append((losses, ts.optimizer.flat.clone(), {k: v.clone() for k, v in st.items()}))
        ts.release_gc()
    (la, pa, sa), (lb, pb, sb) = res
    assert la == lb and torch.equal(pa, pb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_cooperative_batchnorm_modes_train_bit_identically(dev):
    """The cooperative BatchNorm mode only decides WHO computes an item and when (persistent workgroups drawing tickets, one
    workgroup per item, or the per-launch mix of mode 3, the default): the partners' sums are added in slot order either way, so
    three training steps at the headline geometry (64x2048x5, B = 2: fire_blk1's 64x512 planes have 16 partners per channel, all
    launches of mode 3 run one item per workgroup) leave BIT-identical losses, parameters and Adam moments under modes 3, 2 and 1
    (pointseg_modules.py:116-142 in training, trainer.py:263-266)."""
    from deeplio_amd import functional as Fh, ops
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep
    cfg0 = dict(seq=2, overrides=gc.NO_DROP)
    batch = tuple(t.to(dev) for t in gc.make_batch(7, 2, 2, 5, 64, 2048, 50))
    res = []
    try:
        for mode in (3, 2, 1):
            ops.bn_coop_set_mode(mode)
            Fh.manual_seed(11)
            torch.manual_seed(5)
            ts = TrainStep(make_config(**cfg0), (5, 64, 2048), dev, 2)
            gc.fill_state(ts.model, seed=1000)
            losses = [float(ts.step(*batch)) for _ in range(3)]
            ts.check()
            torch.cuda.synchronize()
            st = ts.optimizer._state()
            res.append((losses, ts.optimizer.flat.clone(), {k: v.clone() for k, v in st.items()}))
            ts.release_gc()
            del ts
    finally:
        ops.bn_coop_set_mode(-1)
    assert ops.bn_coop_errors() == 0
    for other in res[1:]:
        assert other[0] == res[0][0] and torch.equal(other[1], res[0][1])
        for k in res[0][2]:
            assert torch.equal(other[2][k], res[0][2][k]), k


def test_lstm_gradient_slots_overwritten_instead_of_zeroed(dev):
    """TrainStep marks the odometry LSTM's gradient slots as overwritten by their producer (FlatOptimizer.set_overwritten:
    functional.LstmStackFn writes a layer's weight gradients in one launch, exactly once per backward pass; the discarded
    direction's slots are never written and stay zero): zero_grad() skips them, the weight-gradient launch does not read them
    back.  Parameters and Adam moments after three steps are BIT-identical to the zero-fill + accumulate path, also when a
    step falls back to the per-direction RNNFn (a batch the layer kernels do not take) -- its first write then replaces the
    stale contents (optimizer.py:4-16, trainer.py:263-266).  Off by default (DLIO_GRAD_OVERWRITE: no measurable gain)."""
    from deeplio_amd import functional as Fh
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep
    g = dict(B=2, S=2, C=5, H=16, W=64, T=7)
    batch = tuple(t.to(dev) for t in gc.make_batch(2000, g['B'], g['S'], g['C'], g['H'], g['W'], g['T']))
    res = []
    for over, fallback_step in ((True, None), (False, None), (True, 1), (False, 1)):
        cfg = make_config(seq=2, overrides=dict(gc.NO_DROP, **{'odom-feat-rnn/hidden-size': 256}))
        os.environ["DLIO_GRAD_OVERWRITE"] = "1" if over else "0"
        try:
            ts = TrainStep(cfg, (g['C'], g['H'], g['W']), dev, g['B'])
        finally:
            os.environ.pop("DLIO_GRAD_OVERWRITE", None)
        gc.fill_state(ts.model, seed=1000)
        skip = getattr(ts.optimizer, "_skip", None)
        assert (skip is not None) == over
        if over:
            assert skip[1] - skip[0] == sum((p.numel() + 15) // 16 * 16 for p in ts.model.odom_feat_net.rnn.parameters())
        losses = []
        for i in range(3):
            Fh._LSTM_LAYER[0] = i != fallback_step
            try:
                losses.append(float(ts.step(*batch)))
            finally:
                Fh._LSTM_LAYER[0] = True
        ts.check()
        torch.cuda.synchronize()
        res.append((losses, ts.optimizer.flat.clone(), ts.optimizer.exp_avg.clone(), ts.optimizer.exp_avg_sq.clone()))
        ts.release_gc()
    for (la, pa, ma, va), (lb, pb, mb, vb) in ((res[0], res[1]), (res[2], res[3])):
        assert la == lb and torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    # (the fallback step computes the same gradients through other kernels: fp32-close losses; Adam's sign-like first updates
    #  turn that into parameter differences of a few 1e-4)
    assert abs(res[2][0][2] - res[0][0][2]) <= 1e-3 * abs(res[0][0][2])
