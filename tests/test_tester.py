"""'next' rows 3 and 4: test-mode path + KITTI trajectory writer, checkpoint layout.
CPU: host logic against the reference golden (tests/golden/tester.npz, written by the reference's
OdomSeqRes) and against torch.optim's own state_dict layout.  GPU: TestStep and resume."""
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "tester.npz"))


def test_odom_seq_res_writes_the_reference_files(tmp_path):
    from deeplio_amd.tester import OdomSeqRes
    seq = OdomSeqRes("2011_10_03", "0027", output_dir=str(tmp_path))
    for i in range(len(GOLD["T_local"])):
        seq.add_local_prediction(GOLD["timestamps"][i], GOLD["loss"][i], GOLD["T_local"][i], GOLD["T_glob"][i])
    gt_name, pred_name = seq.write_to_file()
    assert os.path.basename(gt_name) == "gt_kitti_2011_10_03_0027.txt"
    assert os.path.basename(pred_name) == "pred_kitti_2011_10_03_0027.txt"
    assert open(gt_name).read() == str(GOLD["gt_txt"])            # byte-identical
    assert open(pred_name).read() == str(GOLD["pred_txt"])
    rows = np.loadtxt(pred_name)
    assert rows.shape == (len(GOLD["T_local"]), 12)


def test_local_transform_modes():
    from deeplio_amd.tester import global_transform, local_transform, so3_exp
    from oracle import se3 as ose3
    rng = np.random.default_rng(3)
    for scale in (1e-9, 1e-3, 0.3, 2.5):
        w = (rng.normal(size=3) * scale).astype(np.float32)
        want = ose3.so3_exp(torch.from_numpy(w)).numpy()
        assert np.allclose(so3_exp(w), want, atol=2e-7)
    t, w = rng.normal(size=3).astype(np.float32), (rng.normal(size=3) * 0.1).astype(np.float32)
    gt_t, gt_w = rng.normal(size=3).astype(np.float32), (rng.normal(size=3) * 0.1).astype(np.float32)
    T = local_transform(t, w, gt_t, gt_w, "xq")
    assert T.dtype == np.float64 and np.array_equal(T[:3, 3], t.astype(np.float64)) and np.array_equal(T[3], [0, 0, 0, 1])
    assert np.array_equal(local_transform(t, w, gt_t, gt_w, "x")[:3, :3], so3_exp(gt_w).astype(np.float64))
    assert np.array_equal(local_transform(t, w, gt_t, gt_w, "q")[:3, 3], gt_t.astype(np.float64))
    q = np.array([0.9, 0.1, -0.2, 0.3])
    R = local_transform(t, w, gt_t, q, "gt")[:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1)
    row = np.concatenate([[1, 2, 3], np.arange(9), [0, 0, 0]]).astype(np.float64)
    G = global_transform(row)
    assert np.array_equal(G[:3, 3], [1, 2, 3]) and np.array_equal(G[:3, :3], np.arange(9).reshape(3, 3))


def _mlp():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))


def test_checkpoint_file_layout(tmp_path):
    from deeplio_amd import checkpoint as ck
    m = _mlp()
    p = ck.save_checkpoint({'state_dict': m.state_dict()}, True, str(tmp_path), "cpkt_x")     # trainer.py:439-444
    assert p.endswith("cpkt_x.tar") and os.path.isfile(str(tmp_path / "cpkt_x_best.tar"))
    assert set(torch.load(p, weights_only=False)) == {'state_dict'}
    p = ck.save_checkpoint({'state_dict': m.state_dict()}, False, str(tmp_path), "cpkt_y")
    assert not os.path.exists(str(tmp_path / "cpkt_y_best.tar"))
    with pytest.raises(FileNotFoundError):
        ck.load_training_state(str(tmp_path / "nope.tar"), m)


@pytest.mark.gpu
def test_optimizer_state_is_torch_optim_layout(dev):
    """a torch.optim.Adam state_dict (what the reference's checkpoints hold) loads into the flat
    optimizer, comes back out identical, and the next step agrees with torch's"""
    from deeplio_amd import checkpoint as ck
    from deeplio_amd.optimizer import Adam
    ref, mine = _mlp(), _mlp().to(dev)
    topt = torch.optim.Adam([{'params': ref.parameters()}], lr=2e-3, weight_decay=1e-4)
    for _ in range(3):
        topt.zero_grad()
        ref(torch.ones(4, 5)).square().sum().backward()
        topt.step()
    sd = topt.state_dict()
    fopt = Adam([{'params': mine.parameters()}], lr=1e-3, weight_decay=0.)
    ck.optimizer_from_torch_state(fopt, sd)
    assert fopt.step_count == 3 and fopt.param_groups[0]['lr'] == 2e-3 and fopt.param_groups[0]['weight_decay'] == 1e-4
    back = ck.optimizer_to_torch_state(fopt)
    assert back['param_groups'][0]['params'] == sd['param_groups'][0]['params']
    assert tuple(back['param_groups'][0]['betas']) == tuple(sd['param_groups'][0]['betas'])
    for i, st in sd['state'].items():
        assert torch.equal(back['state'][i]['exp_avg'].cpu(), st['exp_avg'])
        assert torch.equal(back['state'][i]['exp_avg_sq'].cpu(), st['exp_avg_sq'])
        assert float(back['state'][i]['step']) == float(st['step'])
    torch.optim.Adam([{'params': _mlp().parameters()}]).load_state_dict(
        {'state': {i: {k: v.cpu() for k, v in st.items()} for i, st in back['state'].items()},
         'param_groups': back['param_groups']})                                   # torch accepts it
    # one more step on both sides from the same parameters
    mine.load_state_dict(ref.state_dict())
    topt.zero_grad()
    ref(torch.ones(4, 5)).square().sum().backward()
    topt.step()
    fopt.zero_grad()
    mine(torch.ones(4, 5, device=dev)).square().sum().backward()
    fopt.step()
    for a, b in zip(mine.parameters(), ref.parameters()):
        assert float((a.detach().cpu() - b.detach()).abs().max()) < 1e-6


@pytest.mark.gpu
def test_lr_schedule_resumes_across_the_overlay_both_ways(dev):
    """trainer.py:108-114 on resume: optimizer.load_state_dict, then PolynomialLRDecay(optimizer, ..., last_epoch=epoch).
    (a) a checkpoint written through the overlay (FlatOptimizer + misc.PolynomialLRDecay) carries 'initial_lr', loads
    into torch.optim and resumes under a stock torch scheduler; (b) a checkpoint torch.optim + a torch scheduler wrote
    resumes through the overlay on the UNINTERRUPTED schedule (base rate = 'initial_lr', not the decayed 'lr')."""
    import types
    from deeplio_amd.optimizer import create_optimizer
    from deeplio_amd.misc import PolynomialLRDecay
    args = types.SimpleNamespace(lr=1e-3, weight_decay=1e-4, momentum=0.9)

    def flat():
        net = _mlp().to(dev)
        return net, create_optimizer([{'params': net.parameters()}], {'optimizer': 'adam'}, args)

    # the uninterrupted schedule
    _, f0 = flat()
    s0 = PolynomialLRDecay(f0, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    table = []
    for _ in range(30):
        table.append(f0.param_groups[0]['lr'])
        s0.step()
    # (a) overlay writes at epoch 15
    net, f1 = flat()
    s1 = PolynomialLRDecay(f1, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    net(torch.ones(4, 5, device=dev)).square().sum().backward()
    f1.step()
    for _ in range(15):
        s1.step()
    sd = f1.state_dict()
    assert sd['param_groups'][0]['initial_lr'] == 1e-3 and abs(sd['param_groups'][0]['lr'] - table[15]) < 1e-15
    topt = torch.optim.Adam([{'params': _mlp().parameters()}], lr=1e-3, weight_decay=1e-4)
    topt.load_state_dict({'state': {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                                    for i, st in sd['state'].items()}, 'param_groups': sd['param_groups']})
    torch.optim.lr_scheduler.StepLR(topt, 100, last_epoch=15)           # KeyError without 'initial_lr'
    _, f2 = flat()
    f2.load_state_dict(sd)
    s2 = PolynomialLRDecay(f2, max_decay_steps=30, end_learning_rate=1e-6, power=2.0, last_epoch=15)
    assert abs(f2.param_groups[0]['lr'] - table[16]) < 1e-15
    s2.step()
    assert abs(f2.param_groups[0]['lr'] - table[17]) < 1e-15
    # (b) torch.optim + a torch scheduler write at epoch 15 (LambdaLR plays the polynomial: same 'initial_lr' protocol)
    t3 = torch.optim.Adam([{'params': _mlp().parameters()}], lr=1e-3, weight_decay=1e-4)
    sch3 = torch.optim.lr_scheduler.LambdaLR(t3, lambda e: ((1e-3 - 1e-6) * (1 - e / 30) ** 2 + 1e-6) / 1e-3)
    for _ in range(15):
        t3.step()
        sch3.step()
    assert abs(t3.param_groups[0]['lr'] - table[15]) < 1e-12
    _, f3 = flat()
    f3.load_state_dict(t3.state_dict())
    PolynomialLRDecay(f3, max_decay_steps=30, end_learning_rate=1e-6, power=2.0, last_epoch=15)
    assert abs(f3.param_groups[0]['lr'] - table[16]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adam", "sgd", "rmsprop", "adadelta"])
def test_overlay_optimizer_resumes_from_a_torch_optim_checkpoint(dev, tmp_path, kind):
    """the reference's Trainer calls `self.optimizer.state_dict()` when it saves (trainer.py:161) and
    `self.optimizer.load_state_dict(checkpoint['optimizer'])` when it resumes (trainer.py:108).  Through the overlay
    that object is a FlatOptimizer: a checkpoint written by torch.optim (the reference alone) must load into it, what
    it writes must load into torch.optim, and the legacy flat layout of earlier builds still loads."""
    import types
    from deeplio_amd.optimizer import create_optimizer
    mk = {"adam": lambda ps: torch.optim.Adam(ps, lr=2e-3, weight_decay=1e-4),
          "sgd": lambda ps: torch.optim.SGD(ps, lr=2e-3, weight_decay=1e-4, momentum=0.9),
          "rmsprop": lambda ps: torch.optim.RMSprop(ps, lr=2e-3, weight_decay=1e-4),
          "adadelta": lambda ps: torch.optim.Adadelta(ps, lr=2e-3, weight_decay=1e-4)}[kind]
    ref = _mlp()
    topt = mk([{'params': ref.parameters()}])
    for _ in range(3):
        topt.zero_grad()
        ref(torch.ones(4, 5)).square().sum().backward()
        topt.step()
    path = str(tmp_path / "cpkt.tar")
    torch.save({'state_dict': ref.state_dict(), 'optimizer': topt.state_dict()}, path)     # a reference-written file

    mine = _mlp().to(dev)
    args = types.SimpleNamespace(lr=2e-3, weight_decay=1e-4, momentum=0.9)
    fopt = create_optimizer([{'params': mine.parameters()}], {'optimizer': kind}, args)
    ck = torch.load(path, map_location=dev, weights_only=False)
    mine.load_state_dict(ck['state_dict'])
    fopt.load_state_dict(ck['optimizer'])                                                   # trainer.py:108
    for opt, net, x in ((topt, ref, torch.ones(4, 5)), (fopt, mine, torch.ones(4, 5, device=dev))):
        opt.zero_grad()
        net(x).square().sum().backward()
        opt.step()
    for a, b in zip(mine.parameters(), ref.parameters()):
        assert float((a.detach().cpu() - b.detach()).abs().max()) < 2e-6
    # and back: what the overlay writes (trainer.py:161) resumes under torch.optim
    sd = fopt.state_dict()
    assert set(sd) == {'state', 'param_groups'} and all(isinstance(k, int) for k in sd['state'])
    t2 = mk([{'params': _mlp().parameters()}])
    t2.load_state_dict({'state': {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                                  for i, st in sd['state'].items()}, 'param_groups': sd['param_groups']})
    ref2 = t2.param_groups[0]['params']
    with torch.no_grad():
        for p, q in zip(ref2, ref.parameters()):
            p.copy_(q)
    for opt, ps in ((t2, ref2), (topt, list(ref.parameters()))):
        opt.zero_grad()
        for p in ps:
            p.grad = torch.full_like(p, 0.25)
        opt.step()
    for a, b in zip(ref2, ref.parameters()):
        assert float((a.detach() - b.detach()).abs().max()) < 2e-6
    # the flat layout written before round 3
    legacy = {'step': fopt.step_count, 'state': {k: v.clone() for k, v in fopt._state().items()},
              'param_groups': [{'lr': 5e-4, 'weight_decay': 1e-4}]}
    fopt.load_state_dict(legacy)
    assert fopt.param_groups[0]['lr'] == 5e-4


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_teststep_matches_eval_model_and_tester_quaternions(dev):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import golden_common as gc
    from deeplio_amd.config import make_config
    from deeplio_amd.tester import TestStep
    from oracle import model as om, se3 as ose3
    cfg = make_config(lidar="lidar-feat-simple-1", imu="imu-feat-fc", fusion="fusion-layer-cat", odom="odom-feat-fc", seq=1)
    ts = TestStep(cfg, (2, 16, 64), dev, batch_size=2)
    gc.fill_state(ts.model, seed=11)
    batch = tuple(t.to(dev) for t in gc.make_batch(12, 2, 1, 2, 16, 64, 10))
    pt, pw, pp, pq, loss = ts.step(*batch, timed=True)
    assert ts.steps == 1 and ts.inference_time > 0
    omodel = om.get_model((2, 16, 64), cfg)
    gc.fill_state(omodel, seed=11)
    omodel.eval()
    with torch.no_grad():
        opt_, opw = omodel([[batch[0].cpu(), batch[1].cpu()], batch[2].cpu()])
        opp, opq = ose3.se3_to_SE3(opt_, opw, ordering="xyzw")
    for a, b in ((pt, opt_), (pw, opw), (pp, opp), (pq, opq)):
        assert float((a.cpu() - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3)
    assert np.isfinite(float(loss))
    with pytest.raises(ValueError):
        TestStep(make_config(seq=2), (5, 16, 64), dev)


@pytest.mark.gpu
def test_teststep_hipgraph_replay_is_bit_identical(dev):
    """the captured test-mode forward (three streams inside one hipGraph) returns exactly what the
    eager forward returns, also for new inputs copied into the static buffers"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import golden_common as gc
    from deeplio_amd.config import make_config
    from deeplio_amd.tester import TestStep
    ts = TestStep(make_config(seq=1), (5, 64, 256), dev, batch_size=1)
    gc.fill_state(ts.model, seed=3)
    a = tuple(t.to(dev) for t in gc.make_batch(31, 1, 1, 5, 64, 256, 50))
    b = tuple(t.to(dev) for t in gc.make_batch(32, 1, 1, 5, 64, 256, 50))
    ts.capture(*a[:3])
    for batch in (a, b, a):
        with torch.no_grad():
            ref = ts.model([[batch[0], batch[1]], batch[2]])
        got = ts.forward_graph(*batch[:3])
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


@pytest.mark.gpu
def test_resume_from_checkpoint_continues_the_trajectory(dev, tmp_path):
    """train 2 steps, checkpoint, train 2 more; a fresh process-state restored from the files and
    trained 2 steps lands on bit-identical parameters (dropout off)."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import golden_common as gc
    from deeplio_amd import checkpoint as ck
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep
    cfg = make_config(lidar="lidar-feat-simple-1", imu="imu-feat-fc", fusion="fusion-layer-cat", odom="odom-feat-fc", seq=2)
    for k in ("deeplio", "lidar-feat-simple-1", "imu-feat-fc", "odom-feat-fc"):
        if 'dropout' in cfg.get(k, {}):
            cfg[k]['dropout'] = 0.
    batch = tuple(t.to(dev) for t in gc.make_batch(21, 2, 2, 2, 16, 64, 10))

    def fresh():
        ts = TrainStep(cfg, (2, 16, 64), dev, 2)
        gc.fill_state(ts.model, seed=5)
        return ts
    a = fresh()
    for _ in range(2):
        a.step(*batch)
    files = ck.save_training_state(str(tmp_path), 3, a.model, a.criterion, a.optimizer, 0.5, True)
    assert os.path.basename(files[0]) == "cpkt_%s.tar" % a.model.name
    assert all(os.path.isfile(f) and os.path.isfile(f[:-4] + "_best.tar") for f in files)
    assert set(torch.load(files[0], weights_only=False)) == {'epoch', 'state_dict', 'best_acc', 'optimizer', 'criterion'}
    for _ in range(2):
        a.step(*batch)
    b = fresh()
    with torch.no_grad():
        for p in b.model.parameters():
            p.add_(1.0)                       # make sure the load does the work
    epoch, best = ck.load_training_state(files[0], b.model, b.criterion, b.optimizer)
    assert (epoch, best) == (3, 0.5) and b.optimizer.step_count == 2
    for _ in range(2):
        b.step(*batch)
    torch.cuda.synchronize()
    for (k, pa), (_, pb) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(pa, pb), k
    assert torch.equal(a.criterion.sx, b.criterion.sx) and torch.equal(a.criterion.sq, b.criterion.sq)
