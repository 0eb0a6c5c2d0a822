#!/usr/bin/env python3
"""Training-throughput bench of the DeepLIO hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
lidar-feat-pointseg (encoder, fusion=add, bypass simple, dropout .1) + imu-feat-rnn
bi-LSTM 128x2 + fusion-layer-soft + odom-feat-rnn bi-LSTM 1024x2, HWS loss local+global,
Adam(lr 1e-3, wd 1e-4); synthetic 64x2048x5 range-image pairs + 50-step IMU windows,
per-GPU batch 8, S=2 pairs per sample (weak scaling over GPUs).  One "step" = forward,
SE(3) chain, loss, backward, (gradient all-reduce), optimizer step over one batch of
B*S frame pairs already resident in HBM.

Prints ONE JSON line on rank 0 (see the contract in the task description) with two extra
objects: `roofline` for the dominant kernel (hipEvent-timed inside the timed region) and
`cpu_baseline` (the oracle = CPU port of the reference path, timed on this box's host cores on
a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BX3_TFLOPS = 419.5          # fp32-equivalent: dense bf16 MFMA peak (16 x 157.3) / 6 bf16 MFMAs per fp32 product
PEAK_H2_TFLOPS = 838.9           # the same with the two-piece fp16 split: dense fp16 MFMA peak (= bf16) / 3 MFMAs per product
PEAK_HBM_GBS = 8000.0


def synth_batch(seed, B, S, C, H, W, T, device):
    """xyz, normals ~ N(0,1); imu ~ U[0,1); GT f2f_t ~ N(0,.1^2), f2f_w ~ N(0,.01^2),
    f2g_p ~ N(0,1), f2g_q unit-normalised N(0,1)^4."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = torch.randn(B, S, 2, C, H, W, generator=g)
    nrm = torch.randn(B, S, 2, C, H, W, generator=g)
    imu = torch.rand(B, S, T, 6, generator=g)
    f2f = torch.cat([0.1 * torch.randn(B, S, 3, generator=g), 0.01 * torch.randn(B, S, 3, generator=g)], -1)
    q = torch.randn(B, S, 4, generator=g)
    f2g = torch.cat([torch.randn(B, S, 3, generator=g), q / q.norm(dim=-1, keepdim=True)], -1)
    return tuple(t.to(device) for t in (xyz, nrm, imu, f2f, f2g))


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's cores inside a container)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_parity_probe(cfg, C, H, W, T, S, sample_B, device):
    """One un-timed train-mode forward + loss of the HIP path and of the oracle on the SAME weights (the HIP model's
    state_dict loaded into the oracle) and the SAME batch, dropout off (the two sides draw their dropout masks from
    different generators): the relative difference of the two losses -- the timed path checked against the checker at a
    launch size of its own, inside the run that produced the number."""
    import copy
    from deeplio_amd.trainer import TrainStep
    from oracle import model as om
    from oracle import se3 as ose3
    cfg0 = copy.deepcopy(cfg)
    cfg0['deeplio']['dropout'] = 0.
    for k in ('lidar-feat-pointseg', 'lidar-feat-resnet', 'lidar-feat-simple-1', 'lidar-feat-flownet', 'imu-feat-rnn'):
        if k in cfg0 and 'dropout' in cfg0[k]:
            cfg0[k]['dropout'] = 0.
    ts0 = TrainStep(cfg0, (C, H, W), device, sample_B)
    batch = synth_batch(99, sample_B, S, C, H, W, T, "cpu")
    dbatch = tuple(t.to(device) for t in batch)
    with torch.no_grad():
        loss = ts0._tail(ts0.model.forward_features([[dbatch[0], dbatch[1]], dbatch[2]]), dbatch[3], dbatch[4], hook=False)
    ts0.check()
    omodel = om.get_model((C, H, W), cfg0)
    omodel.load_state_dict({k: v.detach().cpu() for k, v in ts0.model.state_dict().items()})
    omodel.train()
    ocrit = om.get_loss_function(cfg0)
    ocrit.load_state_dict({k: v.detach().cpu() for k, v in ts0.criterion.state_dict().items()})
    with torch.no_grad():
        a, b = omodel([[batch[0], batch[1]], batch[2]])
        p2, q2 = ose3.se3_to_SE3(a, b)
        sl = slice(1, ts0.max_glob_seq + 1)
        oloss = ocrit(a, b, p2[:, sl], q2[:, sl], batch[3][:, :, 0:3], batch[3][:, :, 3:], batch[4][:, sl, 0:3],
                      batch[4][:, sl, 3:7])
    lh, lo = float(loss), float(oloss)
    ts0.release_gc()
    return {"loss_hip": lh, "loss_oracle": lo, "loss_rel_err_vs_oracle": abs(lh - lo) / max(abs(lo), 1e-30),
            "probe": "train-mode forward + SE(3) chain + loss, dropout off, same weights and batch, B=%d S=%d" % (sample_B, S)}


def cpu_baseline(cfg, C, H, W, T, S, sample_B, steps, threads=None):
    """The oracle (CPU port of the reference's path) on this box's host cores: full training
    step on a bounded sample of the same workload."""
    from oracle import model as om
    ncores = usable_cores()
    keep = torch.get_num_threads()
    torch.set_num_threads(min(threads, ncores) if threads else ncores)
    try:
        model = om.get_model((C, H, W), cfg)
        model.train()
        crit = om.get_loss_function(cfg)
        opt = om.create_optimizer([{'params': model.parameters()}, {'params': crit.parameters()}], cfg,
                                  lr=1e-3, weight_decay=1e-4)
        batch = synth_batch(99, sample_B, S, C, H, W, T, "cpu")
        om.train_step(model, crit, opt, batch)            # warm-up
        per = []
        for _ in range(steps):
            t0 = time.perf_counter()
            om.train_step(model, crit, opt, batch)
            per.append(time.perf_counter() - t0)
        dt = sum(per) / steps
        used = torch.get_num_threads()
    finally:
        torch.set_num_threads(keep)
    return {"value": round(sample_B * S / dt, 4), "unit": "frame-pairs/s", "cores": used,
            "kind": "port",
            "s_per_step": [round(v, 2) for v in per],
            "sample": "oracle (torch-CPU port of the reference path), same model/loss/Adam step, "
                      "B=%d S=%d 64x2048x%d T=%d, %d timed steps after 1 warm-up, %.2f s/step, %d threads"
                      % (sample_B, S, C, T, steps, dt, used)}


def make_host_batch(seed, B, S, C, H, W, T):
    """what the reference's DataLoader hands DataCombiCreater.process (misc.py:24-63), in pinned memory (its loader runs with
    pin_memory): images [B, S+1, 2C, H, W] (C channels per stream: xyz | normals), imus [B, S, T, 6], gts [B, S+1, 15] =
    rows [position(3), rotation matrix(9), velocity(3)] of a random smooth trajectory"""
    import numpy as np
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = torch.randn(B, S + 1, 2 * C, H, W, generator=g)
    imus = torch.rand(B, S, T, 6, generator=g)
    gts = torch.zeros(B, S + 1, 15)
    rng = np.random.default_rng(seed)
    for b in range(B):
        R, t = np.eye(3), np.zeros(3)
        for f in range(S + 1):
            w = 0.01 * rng.standard_normal(3)
            th = float(np.linalg.norm(w))
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / max(th, 1e-12)
            R = R @ (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K))
            t = t + 0.1 * rng.standard_normal(3)
            gts[b, f] = torch.from_numpy(np.concatenate([t, R.reshape(-1), np.zeros(3)]).astype(np.float32))
    return {'images': images.pin_memory(), 'imus': imus.pin_memory(), 'gts': gts.pin_memory()}


def host_fed_region(ts, B, S, C, H, W, T, device, steps, warmup=4, pool=3):
    """`steps` training steps fed from the HOST (trainer.py:213 + misc.py:24-63): every step takes a pinned host batch (a pool of
    `pool` distinct ones, cycled) -- H2D copies and DataCombiCreater's kernels (pair gather + channel split, ground-truth
    transform) are issued on the 'feed' stream one step AHEAD of the step that consumes them (double-buffered: while step i
    runs, batch i+1 crosses PCIe), the step's stream waits for the batch's event.  -> seconds per step (steady state)"""
    import numpy as np
    from deeplio_amd import functional as Fh
    from deeplio_amd import misc
    comb = np.asarray([[i, i + 1] for i in range(S)])
    feed = Fh.aux_stream(device, "feed")
    dc = misc.DataCombiCreater(comb, device=device, c_split=C)
    hosts = [make_host_batch(4321 + i, B, S, C, H, W, T) for i in range(pool)]
    main = torch.cuda.current_stream()

    def stage(i):
        feed.wait_stream(main)             # (nothing of the step in flight is touched; orders the very first copy)
        with Fh.on_stream(feed):
            dc.process(hosts[i % pool])
            out = (dc.res_imgs, dc.res_normals, dc.res_imu, dc.res_gt_f2f, dc.res_gt_f2g)
            ev = torch.cuda.Event()
            ev.record(feed)
        return out, ev

    nxt = stage(0)
    t0 = None
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        cur, nxt = nxt, None
        main.wait_event(cur[1])
        for t in cur[0]:
            t.record_stream(main)
        nxt = stage(i + 1)                 # the next batch crosses PCIe under this step
        ts.step(*cur[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dc.check()
    h2d = sum(v.numel() * v.element_size() for v in hosts[0].values())
    return dt, h2d


# BASELINE.json configs[2] / [3] / [4] at their per-GPU batch (global batch / DP degree), as short driver-run sub-lines.
# GF / MB per frame pair: SURVEY 8(d) (forward conv FLOPs + RNN; fused-minimum bytes forward, fp32; training = 3 x).
SUB_CONFIGS = (
    dict(key="configs[2]", lidar="lidar-feat-flownet", imu="imu-feat-rnn", fusion="fusion-layer-cat", odom="odom-feat-rnn",
         overrides={'imu-feat-rnn/type': 'gru'}, C=3, S=2, B=4, dtype="f32", gf=172.64 + 0.111, mb=363.0,
         workload="BASELINE configs[2]: lidar-feat-flownet + imu-feat-rnn GRU (bi-GRU-128x2) + fusion-layer-cat + odom-feat-rnn "
                  "bi-LSTM-1024x2, 64x2048x3, T=50, S=2, global bs=16 at DP=4 = per-GPU batch 4 (this line: one rank's share)"),
    dict(key="configs[3]", lidar="lidar-feat-resnet", imu="imu-feat-rnn", fusion="fusion-layer-cat", odom="odom-feat-rnn",
         overrides={'lidar-feat-resnet/fusion': 'cat'}, C=3, S=2, B=4, dtype="f32", gf=178.31 + 0.124, mb=866.0,
         workload="BASELINE configs[3]: lidar-feat-resnet (fusion=cat, build-defined: SURVEY Q1) + imu-feat-rnn bi-LSTM-128x2 + "
                  "fusion-layer-cat + odom-feat-rnn bi-LSTM-1024x2, 64x2048x3 (KITTI geometry, synthetic data), T=50, S=2, "
                  "global bs=32 at DP=8 = per-GPU batch 4 (this line: one rank's share)"),
    dict(key="configs[4]", lidar="lidar-feat-pointseg", imu="imu-feat-rnn", fusion="fusion-layer-soft", odom="odom-feat-rnn",
         overrides={'lidar-feat-pointseg/precision': 'bf16', 'losses/rotation': 'geodesic'}, C=5, S=4, B=8, dtype="bf16",
         gf=36.13 + 0.124, mb=957.0 / 2,
         workload="BASELINE configs[4]: full DeepLIO (PointSeg + bi-LSTM-128x2 + fusion-layer-soft + odom-feat-rnn bi-LSTM-1024x2) "
                  "bf16 mixed precision, geodesic pose loss (HWS), 64x2048x5, T=50, seq_len=4, global bs=64 at DP=8 = per-GPU "
                  "batch 8 (this line: one rank's share)"),
)


def config_line(sc, device, steps, warmup=3):
    """one short training-throughput line for a BASELINE config other than the headline: same TrainStep, same timing rule
    (device syncs on both sides of `steps` steps), the whole-step roofline view on SURVEY 8(d)'s algorithmic figures"""
    import gc as _gc
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep
    cfg = make_config(lidar=sc["lidar"], imu=sc["imu"], fusion=sc["fusion"], odom=sc["odom"], seq=sc["S"],
                      overrides=sc["overrides"])
    torch.manual_seed(20260928)
    ts = TrainStep(cfg, (sc["C"], 64, 2048), device, sc["B"])
    batch = synth_batch(1234, sc["B"], sc["S"], sc["C"], 64, 2048, 50, device)
    ts.check_every = 0                           # (checked here, at the phase boundaries: see main)

    def fell_back():
        try:
            ts.check()
        except RuntimeError as e:
            if "cooperative BatchNorm" not in str(e):
                raise
            print("bench.py: %s: %s" % (sc["key"], e), file=sys.stderr)
            return True
        return False

    for _ in range(warmup):
        ts.step(*batch)
    if fell_back():
        for _ in range(warmup):
            ts.step(*batch)
    # two timed windows of `steps` steps, the line carries the faster one and both figures: a short window is exposed to a
    # one-off host stall (allocator growth after the previous model was freed, a collector pass over its garbage)
    for attempt in range(2):
        windows = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = ts.step(*batch)
            torch.cuda.synchronize()
            windows.append((time.perf_counter() - t0) / steps)
        if not fell_back() or attempt:           # (a fallback inside the windows: timed once more on the two-launch kernels)
            break
    dt = min(windows)
    v = sc["B"] * sc["S"] / dt
    tf, gbs = v * 3 * sc["gf"] * 1e9 / 1e12, v * 3 * sc["mb"] * 1e6 / 1e9
    # fp32 lines: their 3x3 / 3x5 layers form each fp32 product from three fp16 MFMAs (two-piece split; a few layers still from six
    # bf16 MFMAs or on the fp32 MFMA): priced against the two-piece ceiling, the fraction of the fp32-MFMA peak beside it (it can
    # exceed 1: the fp32 matrix cores are not what computes); bf16 line: one bf16 MFMA per product
    peak = PEAK_H2_TFLOPS if sc["dtype"] == "f32" else 2516.8
    out = {"config": sc["key"], "workload": sc["workload"], "value": round(v, 2), "unit": "frame-pairs/s",
           "ms_per_step": round(1e3 * dt, 3), "steps": steps, "warmup": warmup, "dtype": sc["dtype"],
           "windows_ms_per_step": [round(1e3 * w, 3) for w in windows],
           "frame_pairs_per_step": sc["B"] * sc["S"], "loss": float(loss.item()),
           "roofline": {"step": {
               "mfma": {"achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                        "frac_of_fp32_mfma_peak": round(tf / PEAK_F32_MFMA_TFLOPS, 4)},
               "hbm": {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}}}}
    ts.release_gc()
    del ts, batch, loss
    _gc.collect()
    torch.cuda.empty_cache()
    return out


def self_launch(n, dry_run):
    """`python bench.py --gpus N` without torchrun: start N ranks of this very command, one per GPU
    (RANK = LOCAL_RANK = device index, rendezvous on 127.0.0.1), pass rank 0's stdout (the one JSON
    line) through and fail if any rank fails.  The driver's torchrun form sets WORLD_SIZE itself and
    never comes through here."""
    import socket
    import subprocess
    gloo = os.environ.get("DLIO_DIST_BACKEND") == "gloo"      # dry run of the N>1 path: ranks may share a device
    if not dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < 1 or (have < n and not gloo):
            raise SystemExit("bench.py --gpus %d: %d HIP device(s) visible (one rank per GPU; set "
                             "DLIO_DIST_BACKEND=gloo only for a dry run with ranks sharing a device)" % (n, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:                                  # one rank failed: the others would wait forever
                    q.terminate()
        time.sleep(0.05)
    return rc


_WGRAD_FORK_DEFAULT = True


def set_overlap(model, on):
    from deeplio_amd import functional as Fh
    from deeplio_amd import ops
    global _WGRAD_FORK_DEFAULT
    if not on:
        _WGRAD_FORK_DEFAULT = Fh._WGRAD_FORK[0]
        Fh.set_wgrad_stream(False)
        # one stream: a cooperative BatchNorm launch has the chip to itself -- its grid is sized for all of it (the default
        # 5 / 8 leaves room for the neighbours' kernels of the five-stream step; DLIO_BN_COOP_CUS overrides both)
        ops.bn_coop_set_cus(1 << 20)
    else:
        Fh.set_wgrad_stream(_WGRAD_FORK_DEFAULT)
        ops.bn_coop_set_cus(0)
    for m in model.modules():
        if hasattr(m, "two_streams"):
            m.two_streams = on
        if hasattr(m, "side_stream"):
            m.side_stream = on


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # >= 20 timed steps
    ap.add_argument("--warmup", type=int, default=5)     # 5 warm-up steps
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE: 8)")
    ap.add_argument("--seq", type=int, default=None, help="frame pairs per sample (S); default 2 (4 with --dtype bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="samples per CPU-baseline step (default: the timed batch itself)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-baseline steps (BASELINE.md: >= 3) after 1 warm-up")
    ap.add_argument("--cpu8-batch", type=int, default=4,
                    help="samples per step of the extra 8-thread CPU line (BASELINE.md section 2; 0 = skip it)")
    ap.add_argument("--cpu8-steps", type=int, default=2)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short sub-lines for BASELINE configs[2] / [3] / [4] (\"configs\" in the JSON line)")
    ap.add_argument("--config-steps", type=int, default=6, help="timed steps per sub-line (>= 5)")
    ap.add_argument("--host-batch", action="store_true",
                    help="time the steps HOST-FED: a fresh pinned host batch per step through DataCombiCreater on a copy "
                         "stream, double-buffered (the default line reports this figure as well, under \"host_fed\")")
    ap.add_argument("--host-steps", type=int, default=20, help="steps of the host-fed region of the default line (0 = skip)")
    ap.add_argument("--lidar", default="lidar-feat-pointseg", help="informational runs of the other families")
    ap.add_argument("--imu", default="imu-feat-rnn")
    ap.add_argument("--fusion", default="fusion-layer-soft")
    ap.add_argument("--odom", default="odom-feat-rnn")
    ap.add_argument("--channels", type=int, default=5, help="range-image channels per stream (C)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="diagnosis runs: skip the per-family roofline pre-passes AND the extras of the default line "
                         "(host-fed region, sub-lines for configs[2] / [3] / [4])")
    ap.add_argument("--iso-steps", type=int, default=3)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32: the headline (BASELINE configs[1]).  bf16: informational line for BASELINE configs[4] -- "
                         "mixed precision (bf16 activation storage in the PointSeg encoders, fp32 master weights / "
                         "statistics / loss), seq_len 4, geodesic rotation loss, per-GPU batch 8")
    ap.add_argument("--serial", action="store_true",
                    help="diagnosis: whole run with the stream overlap off (one HIP stream; for per-kernel profiles)")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes from here even for --gpus 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: no model, works without a GPU (gloo)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(self_launch(args.gpus, args.dry_run))
    if os.environ.get("DLIO_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DLIO_BENCH_WATCHDOG"]), exit=True)

    from deeplio_amd import dist as ddist
    from deeplio_amd import ops
    from deeplio_amd.config import make_config
    from deeplio_amd.trainer import TrainStep

    world, rank, local = ddist.init()
    if (world > 1 and not args.dry_run and torch.distributed.get_backend() != "nccl"
            and os.environ.get("DLIO_DIST_BACKEND") != torch.distributed.get_backend()):
        # (DLIO_DIST_BACKEND=gloo is the explicit dry run of this path with ranks sharing a device; the line then says so)
        raise SystemExit("bench.py --gpus %d needs the nccl (= RCCL) backend, the process group is %r"
                         % (args.gpus, torch.distributed.get_backend()))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with --nproc-per-node %d, "
                         "or without torchrun (it starts its own ranks)" % (args.gpus, world, args.gpus))
    if args.dry_run:
        # every rank joins the group, one collective, rank 0 reports: proves launcher + rendezvous
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(t)
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": float(t.item()), "value": None}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # one GPU per rank; the modulo only matters for the 2-ranks-on-1-GPU gloo dry run of this path
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)

    bf16 = args.dtype == "bf16"
    if args.seq is None:
        args.seq = 4 if bf16 else 2
    C, H, W, T, S, B = args.channels, 64, 2048, 50, args.seq, args.batch
    cfg = make_config(lidar=args.lidar, imu=args.imu, fusion=args.fusion, odom=args.odom, seq=S)
    if bf16:
        if args.lidar != "lidar-feat-pointseg":
            raise SystemExit("--dtype bf16: the mixed-precision path exists for lidar-feat-pointseg")
        cfg['lidar-feat-pointseg']['precision'] = 'bf16'
        cfg['losses']['rotation'] = 'geodesic'
        args.no_cpu_baseline = True           # the CPU baseline of record belongs to the fp32 headline
    headline = (args.lidar, args.imu, args.fusion, args.odom, C, args.dtype, S) == (
        "lidar-feat-pointseg", "imu-feat-rnn", "fusion-layer-soft", "odom-feat-rnn", 5, "f32", 2)
    torch.manual_seed(20260928)                     # same random-init weights on every rank / run
    ts = TrainStep(cfg, (C, H, W), device, B)
    sync = ddist.GradSync(ts.optimizer.flat, ts.optimizer.grad, ts.optimizer)
    sync.broadcast_parameters()
    ts.set_grad_sync(sync if world > 1 else None)
    batch = synth_batch(1234 + rank, B, S, C, H, W, T, device)
    if args.serial:
        set_overlap(ts.model, False)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Which kernel family dominates is MEASURED first, in two untimed passes of the same step with
    # hipEvent pairs around every launch of every family (convolutions, BatchNorm, pools):
    #   isolated   -- stream overlap off: each family's exclusive chip time;
    #   overlapped -- the real five-stream step: a launch's event-to-event duration includes the share
    #                 of the chip its neighbours on the other streams took.
    # The family with the largest OVERLAPPED time is the dominant one; only it is timed in the timed
    # region (an event pair costs ~1.3 us of stream time; timing all ~400 launches of a step slows it
    # by ~1 ms).
    # The bench checks the device's error words itself, at its phase boundaries: TrainStep's own polling (check_every) would
    # raise from INSIDE a step -- in the middle of a timed region, on one rank only.
    ts.check_every = 0

    def coop_fell_back():
        """ts.check() on every rank; True when ANY rank's cooperative BatchNorm launches hit their spin limit (a shared /
        partitioned device, RCCL's kernels beside them ...): check() has switched that rank to the two-launch kernels -- the line
        says so ("bn_coop": false) and the run goes on (an unattended scaling run must not die here).  Every rank learns it,
        because whatever is repeated afterwards contains collectives and must be repeated by all."""
        bad = 0.0
        try:
            ts.check()
        except RuntimeError as e:
            if "cooperative BatchNorm" not in str(e):
                raise
            print("bench.py: %s" % e, file=sys.stderr)
            bad = 1.0
        return sync.max_over_ranks(bad) > 0.0

    for _ in range(args.warmup):
        ts.step(*batch)
    if coop_fell_back():                         # the warm-up is repeated on the kernels that will be timed
        for _ in range(max(args.warmup, 2)):
            ts.step(*batch)
        ts.check()
    # family -> (profiler kinds, bound, peak, description)
    FAMILIES = {
        # (the launches of fire_blk1-3 run on the two-piece fp16 split -- three MFMAs per fp32 product --, the small maps and the
        #  stems on the three-piece bf16 split -- six: the whole family is priced against the HIGHER ceiling)
        "conv3x3_bx3": ((3,), "mfma", PEAK_H2_TFLOPS, "conv3x3 split-operand MFMA (forward + data gradient; fused Fire expand pair)"),
        "conv2d_1x1": ((2,), "hbm", PEAK_HBM_GBS, "conv2d_1x1 (forward + data gradient, HBM-bound)"),
        "batchnorm": ((6, 7, 8, 9), "hbm", PEAK_HBM_GBS,
                      "batchnorm (train-mode statistics, apply, backward reductions, backward apply; HBM-bound)"),
        "wgrad3x3": ((4,), "mfma", PEAK_H2_TFLOPS, "conv3x3 weight gradient (split-operand MFMA, conv_wgrad3.hip)"),
        "wgrad1x1": ((5,), "hbm", PEAK_HBM_GBS, "conv1x1 weight gradient (HBM-bound)"),
        "pool_se": ((10,), "hbm", PEAK_HBM_GBS, "max-pool with fused SE scale (forward + backward, HBM-bound)"),
        "conv2d_fwd_mfma": ((0,), "mfma", PEAK_F32_MFMA_TFLOPS,
                            "conv2d_fwd_mfma, multi-tap on the fp32 MFMA (forward + data gradient)"),
        "conv2d_wgrad_mfma": ((1,), "mfma", PEAK_F32_MFMA_TFLOPS, "conv2d_wgrad_mfma, stems / strided layers on the fp32 MFMA"),
        # mixed precision: 75 FLOP/B fused is far below the bf16 ridge (315) -> priced in bytes
        "conv_bf16": ((11,), "hbm", PEAK_HBM_GBS, "native-bf16 convolutions (forward, data and weight gradient; HBM-bound)"),
    }
    if not (args.lidar == "lidar-feat-pointseg" and args.dtype == "f32"):
        # FlowNet / ResNet / Simple-1 have no layer on the two-piece kernels (nor has the bf16 path): three-piece ceiling
        for k in ("conv3x3_bx3", "wgrad3x3"):
            FAMILIES[k] = FAMILIES[k][:2] + (PEAK_BX3_TFLOPS,) + FAMILIES[k][3:]
    SUBKIND = {6: "forward statistics", 7: "forward apply", 8: "backward reductions", 9: "backward apply"}
    ALL_KINDS = sorted(k for f in FAMILIES.values() for k in f[0])

    def collect():
        raw = {k: ops.prof_collect(k) for k in ALL_KINDS}
        fam = {}
        for name, (kinds, _, _, _) in FAMILIES.items():
            fam[name] = {key: sum(raw[k][key] for k in kinds) for key in ("ms", "flops", "bytes", "launches")}
            if len(kinds) > 1:
                fam[name]["sub"] = {SUBKIND[k]: raw[k] for k in kinds}
        return fam

    def profiled_pass(steps, overlap):
        if not overlap:
            set_overlap(ts.model, False)
        ops.prof_enable(True)
        ts.step(*batch)                          # creates the event pools
        torch.cuda.synchronize()
        ops.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(*batch)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        ops.prof_enable(False)
        fam = collect()
        if not overlap:
            set_overlap(ts.model, True)
        return fam, ms

    prof_iso = prof_ovl = None
    ms_iso = ms_ovl = None
    DOM = "batchnorm"                            # without the pre-passes (N > 1): the N = 1 choice
    # N > 1: no pre-passes (every rank does the same work either way; the event pools of a pre-pass made
    # the following timed steps 4-7x slower in the two-ranks-on-one-GPU dry run)
    if not args.no_isolated and world == 1 and not args.serial:
        prof_iso, ms_iso = profiled_pass(args.iso_steps, overlap=False)
        prof_ovl, ms_ovl = profiled_pass(args.iso_steps, overlap=True)
        DOM = max(prof_ovl, key=lambda f: prof_ovl[f]["ms"])
        post = int(os.environ.get("DLIO_BENCH_POST", "0"))
        if post & 1:
            ops.prof_release()
        if post & 2:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    if os.environ.get("DLIO_BENCH_DOM"):
        DOM = os.environ["DLIO_BENCH_DOM"]
    prof_on = sum(1 << k for k in FAMILIES[DOM][0]) if os.environ.get("DLIO_BENCH_NOPROF", "0") == "0" else 0
    # at most ~50 timed launches per step in the timed region: event pairs cost stream time and serialise
    # back-to-back launches (timing all 296 BatchNorm launches of a step slowed it from 28.7 to 30.8 ms);
    # a stride coprime with the family's launches per step visits every layer over consecutive steps
    per_step = (prof_ovl[DOM]["launches"] / args.iso_steps) if prof_ovl is not None else 296.0
    stride = 1
    if per_step > 50:
        stride = int(per_step // 50) + 1
        while any(stride % q == 0 and int(per_step) % q == 0 for q in (2, 3, 5, 7, 11, 13)):
            stride += 1
    ops.prof_sample(stride)
    ops.prof_enable(prof_on)
    for attempt in range(2):
        for _ in range(2):                       # back to the plain step; event pool of the chosen family
            ts.step(*batch)
        barrier()
        ops.prof_reset()
        ops.prof_enable(prof_on)
        sync.measure_exposed(world > 1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = ts.step(*batch)
        barrier()
        dt = time.perf_counter() - t0
        ops.prof_enable(False)
        # a fallback INSIDE the timed region: those steps are invalid -- the region is timed once more, on the two-launch kernels
        if not coop_fell_back() or attempt:
            break
    ops.prof_sample(1)
    dt = sync.max_over_ranks(dt)
    exposed = sync.exposed_ms()
    dist_info = sync.describe() if world > 1 else None
    if dist_info is not None:
        if dist_info["world_size"] != args.gpus or dist_info["rank_sum"] != dist_info["rank_sum_expected"]:
            raise SystemExit("the process group did not span %d ranks: %r" % (args.gpus, dist_info))
        dist_info["allreduce_ms_exposed"] = None if exposed is None else round(sync.max_over_ranks(exposed), 3)
    coop_errors = ops.bn_coop_errors()
    if coop_errors:
        raise SystemExit("a cooperative BatchNorm launch hit its spin limit (%d): results invalid" % coop_errors)
    prof_timed = collect()
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * S / (dt / args.steps)

    if rank == 0:
        def rate(v, bound):
            if not v["ms"] > 0:
                return 0.0
            return (v["bytes"] / 1e9 if bound == "hbm" else v["flops"] / 1e12) / (v["ms"] * 1e-3)

        def view(name, v, steps):
            """a kernel family against ITS roofline: MFMA for the multi-tap / weight-gradient kernels
            (algorithmic FLOPs), HBM for the 1x1 / BatchNorm / pool kernels (bytes each launch moves by
            construction: every operand once)"""
            _, bound, peak, desc = FAMILIES[name]
            a = rate(v, bound)
            out = {"kernel": desc, "bound": bound, "achieved": round(a, 2), "peak": peak,
                   "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(a / peak, 4),
                   "launches_per_step": v["launches"] / steps,
                   "avg_launch_ms": round(v["ms"] / max(v["launches"], 1), 5),
                   "ms_per_step_in_kernel": round(v["ms"] / steps, 3)}
            if peak == PEAK_H2_TFLOPS:
                # fp32 products formed from three fp16 MFMAs (two-piece operand split, fp32 accumulation) in the large layers,
                # from six bf16 MFMAs (three-piece split) in the small ones: achieved = ALGORITHMIC fp32 FLOP/s
                out["peak_is"] = ("dense fp16 / bf16 MFMA peak 2516.8 TF/s / 3 MFMAs per fp32 product (two-piece fp16 split: the "
                                  "launches of fire_blk1-3); the three-piece bf16 launches that remain (small maps, stems: "
                                  "6 MFMAs, ceiling 419.5) are priced against the same, higher ceiling")
                out["frac_of_three_piece_ceiling"] = round(a / PEAK_BX3_TFLOPS, 4)
                out["frac_of_fp32_mfma_peak"] = round(a / PEAK_F32_MFMA_TFLOPS, 4)
            elif peak == PEAK_BX3_TFLOPS:
                out["peak_is"] = "bf16 dense MFMA peak 2516.8 TF/s / 6 MFMAs per fp32 product (three-piece bf16 split)"
                out["frac_of_fp32_mfma_peak"] = round(a / PEAK_F32_MFMA_TFLOPS, 4)
            if name == "batchnorm":
                out["bytes_are"] = ("what the launches move by construction: the one-launch kernels (csrc/bn_small.hip: a "
                                    "channel's planes held in registers, partial sums exchanged between workgroups) read "
                                    "every operand once -- backward 3 passes (dy, x, dx; under 'backward apply'), forward on "
                                    "the small maps 2-3 (read, residual, write; under 'forward apply'); the expand layers of "
                                    "fire_blk1-3 take their statistics from the convolution's epilogue and apply in ONE streaming "
                                    "pass (csrc/bn_stream.hip: read, residual, write -- or, in front of SELayer + pool, write only "
                                    "the pooled tensor); the two-launch kernels that remain (squeeze layers, the stem behind its "
                                    "pool) 1 + 2-3 and 2 + 3; SURVEY 8(d)'s fused minimum counts 0 bytes for BatchNorm, i.e. all "
                                    "of this is overhead relative to it")
                if "sub" in v:
                    out["kernels"] = {k: {"GB/s": round(rate(sv, "hbm"), 1), "ms_per_step": round(sv["ms"] / steps, 3),
                                          "launches_per_step": sv["launches"] / steps} for k, sv in v["sub"].items()}
            return out

        # PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction of
        # the guide applied): collected by tools/pmc_traffic.py, committed under profiles/
        pmc = None
        for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            path = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(path) and B == 8 and S == 2 and headline:
                with open(path) as f:
                    pmc = (fn, json.load(f))
                break

        def traffic(name, v, steps):
            t = pmc[1].get(name) if pmc else None
            if not t or not v["launches"]:
                return None
            return round(t["hbm_bytes_per_step_corrected"] / (v["launches"] / steps))

        p = prof_timed[DOM]
        roofline = view(DOM, p, args.steps)
        if stride > 1:
            # sums over the sampled launches; scale the per-step figures back to all launches of the family
            roofline["sampled"] = "1 launch in %d timed (achieved / avg_launch_ms are over the sampled launches)" % stride
            roofline["launches_per_step"] = per_step
            roofline["ms_per_step_in_kernel"] = round(roofline["avg_launch_ms"] * per_step, 3)
            roofline.pop("kernels", None)
            if prof_ovl is not None and "sub" in prof_ovl[DOM]:    # per-kernel split: from the overlapped pre-pass
                roofline["kernels"] = view(DOM, prof_ovl[DOM], args.iso_steps)["kernels"]
        roofline["traffic"] = traffic(DOM, {"launches": per_step * args.steps} if stride > 1 else p, args.steps)
        if roofline["traffic"] is not None:
            roofline["traffic_unit"] = "HBM bytes per launch (avg), from profiles/%s" % pmc[0]
        roofline["note"] = ("timed region: kernels of 5 concurrent HIP streams (2 encoders, their weight-gradient "
                            "companions, IMU branch) share the chip, so per-launch durations include the neighbours' "
                            "share; only the family with the largest OVERLAPPED time is timed there (an event pair "
                            "costs stream time).  'other' = every family in an untimed pass of the same overlapped "
                            "step with all launches timed; 'isolated' = the same with the stream overlap off.  The "
                            "cooperative BatchNorm launches of fire_blk2 / blk3 run one item per workgroup (mode 3, DESIGN 9): "
                            "their workgroups come and go between the neighbours' instead of holding 3 / 8 of the chip for the "
                            "launch's length -- each launch takes longer INSIDE the overlapped step (frac 0.27 -> 0.245) while "
                            "the step and the family alone get faster (isolated.frac 0.45 -> 0.47)")
        if prof_ovl is not None:
            other = {}
            for name, v in prof_ovl.items():
                if name == DOM or not v["launches"]:
                    continue
                other[name] = view(name, v, args.iso_steps)
                other[name]["traffic"] = traffic(name, v, args.iso_steps)
            roofline["other"] = other
            roofline["other_pass"] = {"what": "untimed overlapped pass, every family timed (%d steps)" % args.iso_steps,
                                      "ms_per_step": round(ms_ovl, 3),
                                      "dominant_there": view(DOM, prof_ovl[DOM], args.iso_steps)}
        # whole-step view on SURVEY 8(d)'s algorithmic figures: 3 x (36.13 GF conv + 0.124 GF RNN) and
        # 3 x 957 MB per frame pair
        pairs_per_s = value
        if headline:
            roofline["step"] = {
                "mfma": {"achieved": round(pairs_per_s * 3 * 36.254e9 / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(pairs_per_s * 3 * 36.254e9 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
                "hbm": {"achieved": round(pairs_per_s * 3 * 957e6 / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(pairs_per_s * 3 * 957e6 / 1e9 / PEAK_HBM_GBS, 4)}}
            if pmc and "all" in pmc[1]:
                tb = pmc[1]["all"]["hbm_bytes_per_step_corrected"]
                roofline["step"]["hbm_measured"] = {
                    "bytes_per_step": round(tb), "achieved": round(tb / (ms_per_step * 1e-3) / 1e9, 1),
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(tb / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                    "source": "profiles/%s (PMC fabric bytes of one step)" % pmc[0]}
        if prof_iso is not None:
            iso = {"what": "same kernels, same step, stream overlap off (%d untimed steps)" % args.iso_steps,
                   "ms_per_step": round(ms_iso, 3)}
            iso.update(view(DOM, prof_iso[DOM], args.iso_steps))
            iso["other"] = {name: view(name, v, args.iso_steps) for name, v in prof_iso.items()
                            if name != DOM and v["launches"]}
            roofline["isolated"] = iso
        out = {
            "metric": "frame-pairs/sec training, 64x2048x5 range-img + 50-step IMU, bs=8, 1/2/4/8 GPU",
            "value": round(value, 3), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "dtype_note": ("bf16 storage of the PointSeg encoders' activations and activation gradients, one bf16 MFMA per "
                           "product with fp32 accumulation; fp32 master weights, weight gradients, BatchNorm statistics, stem "
                           "convolution, RNNs, heads, SE(3) chain, loss and Adam (deeplio_amd/mixed.py; "
                           "tests/test_gpu_mixed.py)") if bf16 else
                          ("fp32 tensors, accumulation and results throughout; the 3x3 convolutions (forward, data and weight gradient) "
                           "and the large 1x1 layers form each fp32 product on the 16-bit matrix cores from operand splits: three "
                           "fp16 MFMAs over a two-piece split of x 2^k (k from the tensor's largest magnitude or a bound on it) in "
                           "the layers of fire_blk1-3, six bf16 MFMAs over a three-piece split elsewhere -- error vs fp64 = the fp32 "
                           "MFMA's in both (1.5-5e-7 of the result's scale; tests/test_gpu_ops.py::test_conv3x3_split_bf16_matches_fp64, "
                           "::test_conv3x3_two_piece_fp16_data_gradient, ::test_conv3x3_two_piece_fp16_weight_gradient, "
                           "::test_conv1x1_two_piece_fp16_data_gradient)"),
            "config": {"workload": ("BASELINE configs[1]: lidar-feat-pointseg(add)+imu-feat-rnn bi-LSTM-128x2"
                                    "+fusion-layer-soft+odom-feat-rnn bi-LSTM-1024x2, HWS local+global, Adam; "
                                    "64x2048x5, T=50, S=%d, per-GPU batch %d" % (S, B)) if headline else
                                   ("informational (not the headline): BASELINE configs[4] -- full DeepLIO (PointSeg + bi-LSTM "
                                    "+ soft fusion + odometry bi-LSTM) bf16 mixed precision, geodesic rotation loss (HWS), Adam; "
                                    "64x2048x%d, T=50, seq_len %d, per-GPU batch %d (global batch 64 at DP=8)" % (C, S, B)) if bf16 else
                                   ("informational (not the headline config): %s+%s+%s+%s, 64x2048x%d, T=50, S=%d, "
                                    "per-GPU batch %d" % (args.lidar, args.imu, args.fusion, args.odom, C, S, B)),
                       "global_batch": world * B, "frame_pairs_per_step": world * B * S,
                       "parallelism": "dp%d" % world, "loss": float(loss.item()),
                       # the cooperative one-launch BatchNorm kernels were in use for the whole timed region (False: a launch
                       # hit its spin limit during warm-up and the run fell back to the two-launch kernels)
                       "bn_coop": bool(ops._BN_COOP[0]),
                       # functional.assign_streams: the step's four heavy streams found four different hardware queues (4)
                       "stream_queues": __import__("deeplio_amd.functional", fromlist=["x"]).ASSIGN_REPORT.get(
                           device.index if device.index is not None else 0)},
            "roofline": roofline,
        }
        if world == 1 and headline and (args.host_batch or (args.host_steps > 0 and not args.no_isolated)) and not args.serial:
            hs = args.steps if args.host_batch else args.host_steps
            hdt, h2d = host_fed_region(ts, B, S, C, H, W, T, device, hs)
            if coop_fell_back():                 # (the region once more, on the kernels the run fell back to)
                hdt, h2d = host_fed_region(ts, B, S, C, H, W, T, device, hs)
                ts.check()
            out["host_fed"] = {
                "value": round(B * S / hdt, 3), "unit": "frame-pairs/s", "ms_per_step": round(1e3 * hdt, 3), "steps": hs,
                "vs_device_resident": round((B * S / hdt) / value, 4), "h2d_bytes_per_step": h2d,
                "what": "the same step fed from the HOST: a pinned host batch per step (pool of 3 distinct ones) -> H2D + "
                        "DataCombiCreater (pair gather + channel split, ground-truth transform; misc.py:24-63) on a copy "
                        "stream one step ahead (double-buffered), the step waits for the batch's event; `value` above is "
                        "the device-resident figure the contract asks for"}
        if world == 1 and headline and not args.no_configs and not args.no_isolated and not args.serial:
            out["configs"] = [config_line(sc, device, max(5, args.config_steps)) for sc in SUB_CONFIGS]
        if dist_info is not None:
            # what the N > 1 line rests on: the backend and the world size read back from the process group, a collective only
            # N ranks answer correctly, the part of the gradient exchange the step waited for, the overlap switch
            out["dist"] = dist_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, C, H, W, T, S, args.cpu_batch, args.cpu_steps)
            if args.cpu8_batch > 0 and usable_cores() > 8:
                # the n = 8 thread line beside the all-cores one (comparable with SURVEY's 8-core probe)
                out["cpu_baseline"]["threads_8"] = cpu_baseline(cfg, C, H, W, T, S, args.cpu8_batch, args.cpu8_steps, threads=8)
            out["cpu_baseline"].update(cpu_parity_probe(cfg, C, H, W, T, S, args.cpu_batch, device))
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
