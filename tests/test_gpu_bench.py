"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys, the roofline
and (when asked) the CPU-baseline objects; run end to end in a subprocess at the headline shape."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract(dev):
    d = _run("--no-cpu-baseline", "--iso-steps", "1", "--no-configs", "--host-steps", "0")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "frame-pairs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frame pairs per step / time per step
    assert abs(d["value"] - d["config"]["frame_pairs_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert 0.0 < r["frac"] < 1.0
    # live event timing of the family with the largest OVERLAPPED time (launches per step at the headline
    # shape)
    per_step = {"conv3x3 split-bf16": 50.0, "conv3x3 weight gradient": 24.0, "conv1x1 weight gradient": 48.0,
                "conv2d_wgrad": 2.0, "conv2d_1x1": 72.0, "conv2d_fwd_mfma": 2.0, "batchnorm": 134.0, "max-pool": 20.0}
    fam = next(k for k in per_step if r["kernel"].startswith(k))
    # (conv3x3: 24 fused expand pairs + 24 data gradients + 2 stems; 1x1: squeeze forward / data gradient + expand1x1 data
    #  gradient, its forward rides in the fused pair.  BatchNorm: squeeze 24 x (statistics + split apply) forward and 24
    #  one-launch backward; expand: 18 finalising launches (every fused block of fire_blk1-3 and fire_blk5's first takes its
    #  sums out of the expand launch's epilogue) + 8 streaming applies (6 of them pooling as well) + 6 one-launch forward on
    #  the small maps, 24 one-launch backward; stem: statistics + the two pool-folded backward launches per encoder = 134;
    #  max-pool / SELayer: 14 + the 6 scale passes over pooled tensors)
    want = (per_step[fam],)
    assert r["launches_per_step"] in want and r["avg_launch_ms"] > 0
    # every family, BatchNorm and the pools included, is a candidate: measured in the overlapped pre-pass
    other = r["other"]
    names = set(other) | {next(n for n in ("conv3x3_bx3", "conv2d_1x1", "batchnorm", "wgrad3x3", "wgrad1x1", "pool_se",
                                           "conv2d_fwd_mfma", "conv2d_wgrad_mfma")
                               if n not in other)}
    # (the fp32 multi-tap family is empty at the headline shape since the PointSeg stem runs on the split-bf16 kernel)
    assert names - {"conv2d_fwd_mfma"} == {"conv3x3_bx3", "conv2d_1x1", "batchnorm", "wgrad3x3", "wgrad1x1", "pool_se",
                                           "conv2d_wgrad_mfma"}
    dom_ovl = r["other_pass"]["dominant_there"]["ms_per_step_in_kernel"]
    assert all(dom_ovl >= v["ms_per_step_in_kernel"] for v in other.values())     # dominant on OVERLAPPED time
    bn = other.get("batchnorm") or r
    assert bn["bound"] == "hbm" and set(bn["kernels"]) == {"forward statistics", "forward apply", "backward reductions",
                                                           "backward apply"}
    assert "traffic" in bn
    iso = r["isolated"]
    assert iso["launches_per_step"] in want
    assert iso["ms_per_step_in_kernel"] < r["other_pass"]["dominant_there"]["ms_per_step_in_kernel"]   # alone: faster
    assert len(iso["other"]) in (6, 7)        # the other families (the fp32 multi-tap one is empty at the headline shape)


def test_bench_cpu_baseline_object(dev):
    d = _run("--no-isolated", "--cpu-steps", "2", "--cpu-batch", "1", "--cpu8-batch", "1", "--cpu8-steps", "1", "--no-configs",
             "--host-steps", "0")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1 and c["value"] > 0
    assert len(c["s_per_step"]) == 2 and "2 timed steps" in c["sample"]
    if c["cores"] > 8:                     # the n = 8 thread line beside the all-cores one
        c8 = c["threads_8"]
        assert c8["cores"] == 8 and c8["kind"] == "port" and c8["value"] > 0
    assert c["loss_rel_err_vs_oracle"] < 1e-4


def test_bench_sub_lines_and_host_fed(dev):
    """the default line also carries (a) short driver-run lines for BASELINE configs[2] / [3] / [4] at their per-GPU batch --
    FlowNet + GRU + fusion-layer-cat, ResNet(cat) + bi-LSTM, bf16 PointSeg S = 4 with the geodesic loss -- and (b) the
    host-fed figure: the same step with a pinned host batch per step through DataCombiCreater on the copy stream"""
    d = _run("--no-cpu-baseline", "--iso-steps", "1", "--host-steps", "4", "--config-steps", "5")
    subs = d["configs"]
    assert [s["config"] for s in subs] == ["configs[2]", "configs[3]", "configs[4]"]
    for s, (fp, dt, words) in zip(subs, ((8, "f32", ("lidar-feat-flownet", "GRU", "fusion-layer-cat")),
                                         (8, "f32", ("lidar-feat-resnet", "fusion=cat", "bi-LSTM")),
                                         (32, "bf16", ("bf16", "geodesic", "seq_len=4")))):
        assert s["steps"] >= 5 and s["frame_pairs_per_step"] == fp and s["dtype"] == dt
        assert all(w in s["workload"] for w in words), s["workload"]
        assert abs(s["value"] - fp / (s["ms_per_step"] * 1e-3)) <= 1e-2 * s["value"]
        r = s["roofline"]["step"]
        assert 0 < r["mfma"]["frac"] < 1 and 0 < r["hbm"]["frac"] < 1
        assert s["loss"] == s["loss"] and abs(s["loss"]) < 1e6          # finite
    h = d["host_fed"]
    assert h["steps"] == 4 and h["value"] > 0 and h["h2d_bytes_per_step"] > 100e6
    assert abs(h["vs_device_resident"] - h["value"] / d["value"]) <= 1e-3


def _run_env(env, *extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launch_one_rank(dev):
    """--gpus 1 through the launcher that `python bench.py --gpus N` uses when no torchrun set WORLD_SIZE"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _run_env(env, "--gpus", "1", "--spawn", "--no-cpu-baseline", "--no-isolated", "--no-configs", "--host-steps", "0")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "dp1"


def test_bench_self_launch_two_ranks_share_the_gpu(dev):
    """`python bench.py --gpus 2` with no torchrun: two ranks are spawned, the gradient exchange runs
    (gloo transport: RCCL refuses two ranks on one device, and the test box has one GPU), rank 0
    prints one line with n_gpus = 2 and the whole-job rate over both ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["DLIO_DIST_BACKEND"] = "gloo"
    d = _run_env(env, "--gpus", "2", "--batch", "2", "--no-cpu-baseline", "--no-isolated")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["frame_pairs_per_step"] == 8
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    # the line carries what it rests on: backend and world size read back from the process group, a collective only two
    # ranks answer with 3, the exposed part of the gradient exchange, the two buckets
    ds = d["dist"]
    assert ds["backend"] == "gloo" and ds["world_size"] == 2 and ds["rank_sum"] == 3.0 == ds["rank_sum_expected"]
    assert ds["allreduce_ms_exposed"] is not None and ds["allreduce_ms_exposed"] >= 0 and len(ds["buckets_bytes"]) in (1, 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices (RCCL refuses two ranks on one)")
def test_bench_two_ranks_over_rccl(dev):
    """`python bench.py --gpus 2` on two devices: the default backend is nccl (= RCCL), anything else is refused; the
    line proves the process group spanned two ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DLIO_DIST_BACKEND")}
    d = _run_env(env, "--gpus", "2", "--batch", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-isolated")
    ds = d["dist"]
    assert d["n_gpus"] == 2 and ds["backend"] == "nccl" and ds["world_size"] == 2 and ds["rank_sum"] == 3.0
    assert ds["allreduce_ms_exposed"] is not None and d["value"] > 0
