"""Data-parallel parity ON THE GPU: two ranks share the one GPU of the test box and talk
over gloo (RCCL refuses two ranks on one device; the collective layer is torch.distributed either
way).  With synchronised BatchNorm statistics a global batch of 4 split 2+2 must train exactly like
the same 4 samples in one process: loss, every gradient, BN running statistics."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu

SHAPE, T, GB = (5, 64, 256), 50, 4

WORKER = r'''
import os, sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(gold)r)
import golden_common as gc
from deeplio_amd import dist as ddist
from deeplio_amd.trainer import TrainStep
sys.path.insert(0, %(here)r)
from test_gpu_dist import make_cfg, SHAPE, T, GB
world, rank, local = ddist.init("gloo")
dev = torch.device("cuda", 0)
ts = TrainStep(make_cfg(), SHAPE, dev, GB // world)
gc.fill_state(ts.model, seed=77)
sync = ddist.GradSync(ts.optimizer.flat, ts.optimizer.grad, ts.optimizer)
sync.broadcast_parameters()
sync.enable_sync_bn(%(sync_bn)s)
ts.set_grad_sync(sync)
full = gc.make_batch(500, GB, 2, SHAPE[0], SHAPE[1], SHAPE[2], T)
per = GB // world
batch = tuple(t[rank * per:(rank + 1) * per].contiguous().to(dev) for t in full)
loss = ts.step(*batch)
torch.cuda.synchronize()
lt = loss.detach().clone().reshape(1).cpu()
torch.distributed.all_reduce(lt)
if rank == 0:
    bufs = {k: v.detach().cpu() for k, v in ts.model.named_buffers() if k.endswith("running_mean") or k.endswith("running_var")}
    torch.save({"loss": float(lt) / world, "grad": (ts.optimizer.grad / world).cpu(), "bufs": bufs,
                "tail": sync.tail_lo}, %(out)r)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


def make_cfg():
    from deeplio_amd.config import make_config
    return make_config(seq=2, overrides={"deeplio/dropout": 0., "lidar-feat-pointseg/dropout": 0.,
                                         "imu-feat-rnn/dropout": 0.})


def _run_two_ranks(tmp_path, sync_bn, port, pin=True):
    out = str(tmp_path / ("dp_%d.pt" % sync_bn))
    script = tmp_path / ("worker_%d.py" % sync_bn)
    script.write_text(WORKER % dict(root=ROOT, gold=os.path.join(HERE, "golden"), here=HERE, out=out,
                                    sync_bn="True" if sync_bn else "False"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    if pin:
        env["DLIO_BX3_1X1_MIN"] = SAME_KERNELS
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return torch.load(out, weights_only=False)


# The 1x1 routing looks at the pixel count of the launch (functional._use_bx3), which halves per rank: a layer may
# run on the split-bf16 kernel in one process and on the fp32-MFMA kernel in two.  Both are fp32-accurate, but their
# roundings differ and the encoder gradients amplify 1e-7 in an activation to 1e-3 (ReLU flips) -- the
# comparison WITHOUT synchronised statistics pins the kernels on both sides.  With synchronised statistics (the parity
# mode) GradSync.enable_sync_bn routes by the global launch size, so the production defaults are what is compared.
SAME_KERNELS = "16,16,1,1"


def _single_process(dev, pin=True):
    import golden_common as gc
    from deeplio_amd import functional as Fh
    from deeplio_amd.trainer import TrainStep
    saved = list(Fh._BX3_1X1_MIN)
    saved_small, saved_stats, saved_h2 = Fh._BN_SMALL[0], Fh._FIRE_STATS[0], Fh._FIRE_H2[0]
    if pin:
        Fh._BX3_1X1_MIN[:] = [int(v) for v in SAME_KERNELS.split(",")]
    else:
        # synchronised statistics need the partial sums BETWEEN two launches: the data-parallel side runs the two-launch
        # BatchNorm kernels, so the single process is compared on them too (the one-launch kernels of bn_small.hip give the
        # same statistics to 1e-7, which the ReLU masks amplify to 1e-3 in the encoder gradients like any other rounding)
        Fh._BN_SMALL[0] = False
        Fh._FIRE_STATS[0] = False           # (and the statistics pass instead of the expand launch's tile sums)
        Fh._FIRE_H2[0] = False              # (and the three-piece planes: the two-piece ones need local batch statistics)
    try:
        ts = TrainStep(make_cfg(), SHAPE, dev, GB)
        gc.fill_state(ts.model, seed=77)
        batch = tuple(t.to(dev) for t in gc.make_batch(500, GB, 2, SHAPE[0], SHAPE[1], SHAPE[2], T))
        loss = ts.step(*batch)
        torch.cuda.synchronize()
    finally:
        Fh._BX3_1X1_MIN[:] = saved
        Fh._BN_SMALL[0], Fh._FIRE_STATS[0], Fh._FIRE_H2[0] = saved_small, saved_stats, saved_h2
    bufs = {k: v.detach().cpu() for k, v in ts.model.named_buffers() if k.endswith("running_mean") or k.endswith("running_var")}
    return float(loss), ts.optimizer.grad.cpu().clone(), bufs, ts


def test_two_ranks_with_sync_bn_equal_one_process(dev, tmp_path):
    loss1, grad1, bufs1, ts = _single_process(dev, pin=False)
    dp = _run_two_ranks(tmp_path, True, 29541, pin=False)
    assert dp["tail"] is not None                       # the overlapped tail bucket was in use
    assert abs(dp["loss"] - loss1) <= 1e-5 * abs(loss1), (dp["loss"], loss1)
    for k, v in bufs1.items():
        assert float((dp["bufs"][k] - v).abs().max()) <= 1e-5 * max(float(v.abs().max()), 1e-3), k
    # gradients, per parameter tensor (two HIP runs of the same math, only the reduction trees differ)
    errs = []
    for p, o in zip(ts.optimizer.params, ts.optimizer.offsets):
        a, b = dp["grad"][o:o + p.numel()].double(), grad1[o:o + p.numel()].double()
        if float(b.abs().max()) < 1e-5 * float(grad1.abs().max()):
            continue
        errs.append(float((a - b).norm()) / max(float(b.norm()), 1e-30))
    errs = np.asarray(errs)
    print("sync-BN DP=2 vs 1 process: grad rel-L2 median %.2e max %.2e" % (np.median(errs), errs.max()))
    assert np.median(errs) <= 1e-5 and errs.max() <= 1e-4        # measured: 6e-8 / 7e-7


def test_two_ranks_without_sync_bn_differ_only_through_batchnorm(dev, tmp_path):
    """per-replica statistics (the throughput configuration) are NOT the single-device result: the
    switch matters, and everything that does not pass through a BatchNorm still agrees"""
    loss1, grad1, bufs1, ts = _single_process(dev)
    dp = _run_two_ranks(tmp_path, False, 29542)
    k = next(iter(bufs1))
    assert float((dp["bufs"][k] - bufs1[k]).abs().max()) > 1e-6 * float(bufs1[k].abs().max())
    assert abs(dp["loss"] - loss1) <= 5e-2 * abs(loss1)          # same model, slightly different statistics


RCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(gold)r)
import golden_common as gc
from deeplio_amd import dist as ddist, ops
from deeplio_amd.trainer import TrainStep
sys.path.insert(0, %(here)r)
from test_gpu_dist import make_cfg, SHAPE, T, GB
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
res = {}
for mode in ("plain", "rccl"):
    ts = TrainStep(make_cfg(), SHAPE, dev, GB)
    gc.fill_state(ts.model, seed=77)
    if mode == "rccl":
        sync = ddist.GradSync(ts.optimizer.flat, ts.optimizer.grad, ts.optimizer)
        sync.world = 2                   # open every world > 1 branch; the group itself has one rank,
        sync.broadcast_parameters()      # so each collective is the identity and grad_scale stays 1
        ops.set_sync_bn(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), 1)
        ts.set_grad_sync(sync)
        assert sync.tail_lo is not None
    batch = tuple(t.to(dev) for t in gc.make_batch(500, GB, 2, SHAPE[0], SHAPE[1], SHAPE[2], T))
    losses = [float(ts.step(*batch)) for _ in range(2)]
    torch.cuda.synchronize()
    if mode == "rccl":
        assert sync.max_over_ranks(1.5) == 1.5
        dist.barrier()
        ops.set_sync_bn(None, 1)
    res[mode] = {"loss": losses, "grad": ts.optimizer.grad.cpu().clone(), "flat": ts.optimizer.flat.detach().cpu().clone()}
torch.save(res, %(out)r)
dist.destroy_process_group()
'''


def test_collective_call_sites_run_through_rccl(dev, tmp_path):
    """The N>1 path cannot be launched on a 1-GPU box over RCCL (it refuses two ranks on one device),
    but every collective call site can: a one-rank RCCL group with the world > 1 branches forced open
    (parameter broadcast, async tail-bucket all-reduce from the autograd hook, head-bucket all-reduce,
    SyncBN partial sums, max-over-ranks, barrier).  Each collective is then the identity, so two
    optimizer steps must be bit-identical to the non-distributed run."""
    out = str(tmp_path / "rccl.pt")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % dict(root=ROOT, gold=os.path.join(HERE, "golden"), here=HERE, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    res = torch.load(out, weights_only=False)
    assert res["plain"]["loss"] == res["rccl"]["loss"]
    assert torch.equal(res["plain"]["grad"], res["rccl"]["grad"])
    assert torch.equal(res["plain"]["flat"], res["rccl"]["flat"])
