"""Data-parallel execution of the hot path: one process per GPU, torch.distributed with the
"nccl" backend (= RCCL on ROCm) over xGMI.  The reference is single-device (no DataParallel,
no torch.distributed anywhere); batches of frame-pair sequences are independent, so the only
exchange is the gradient all-reduce, issued as ONE collective over the flat gradient buffer
(165 MB fp32 for the PointSeg model).  The exchange is split in two buckets along the
backward order: the odometry net + heads + loss weights (the tail of the flat buffer, 143 of the
165 MB) are complete as soon as backward reaches the fusion output, so their all-reduce is
started there (asynchronously, from an autograd hook) and runs over xGMI underneath the ~20 ms
of encoder backward; only the remaining 22 MB are reduced after backward (~0.3 ms exposed
instead of ~1.9 ms).

BatchNorm statistics are per replica in the throughput configuration; `GradSync.enable_sync_bn()`
switches to synchronised statistics (exact equivalence with one device at the same global batch,
tested with two ranks).  The averaged gradient is exact for everything else.  Works on CPU tensors with the "gloo" backend, which is how the
-m "not gpu" tests cover the world_size > 1 path.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (MASTER_ADDR,
    MASTER_PORT, RANK, WORLD_SIZE).  Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("DLIO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                raise RuntimeError("rank %d (LOCAL_RANK %d) has no GPU: %d HIP device(s) visible -- RCCL needs one "
                                   "device per rank" % (rank, local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_batch(global_batch, world, rank):
    """contiguous split of sample indices [0, global_batch) over ranks (B/world each)"""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return range(rank * per, (rank + 1) * per)


class GradSync:
    """Keeps replicas identical: broadcast of the flat parameter buffer at start, one
    all-reduce(sum) of the flat gradient buffer per step; the optimizer divides by world
    through its `grad_scale` (no extra pass over the gradients)."""

    def __init__(self, flat_param, flat_grad, optimizer=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.flat_param, self.flat_grad = flat_param, flat_grad
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        self.tail_lo = None          # element offset where the early bucket starts
        self._tail_work = None
        self._timing = None          # (start events, end events) of all_reduce_grads when measure_exposed(True)
        # (the cooperative BatchNorm launches need no co-residency with RCCL's persistent kernels -- ticket dispenser,
        #  csrc/bn_small.hip -- and their default grid leaves 3 / 8 of the chip's wave slots alone: nothing to set up here)

    def enable_sync_bn(self, on=True):
        """synchronised BatchNorm statistics: the per-channel partial sums of every train-mode BN
        (forward: sum x, sum x^2; backward: sum g, sum g*xhat) are all-reduced between the reduce and
        the apply kernel, so a global batch split over replicas trains exactly like the same batch on
        one device (the reference is single-device).  Costs two small collectives per BN layer and
        step: a parity switch, off for throughput runs."""
        from . import functional as Fh
        from . import ops
        if on and self.world > 1:
            ops.set_sync_bn(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), self.world)
            Fh._ROUTE_WORLD[0] = self.world       # kernel routing by the global launch size: same kernels as one process
        else:
            ops.set_sync_bn(None, 1)
            Fh._ROUTE_WORLD[0] = 1

    def set_tail(self, lo):
        """gradients [lo:] of the flat buffer are final when `reduce_tail_async` is called"""
        if os.environ.get("DLIO_DP_OVERLAP", "1") == "0" or lo is None or lo <= 0 or lo >= self.flat_grad.numel():
            self.tail_lo = None
        else:
            self.tail_lo = int(lo)

    def reduce_tail_async(self):
        """start the all-reduce of the tail bucket (called from the autograd hook on the odometry
        net's input: every kernel that wrote these gradients ran on the hook's current stream or
        on its weight-gradient companion)"""
        if self.world > 1 and self.tail_lo is not None and self._tail_work is None:
            if self.flat_grad.is_cuda:
                from .functional import aux_stream, wgrad_stream_of
                # Issued as a SYNCHRONOUS collective under a stream of our own: current process groups launch such a
                # collective on the current stream, older ones on an internal stream the current one then waits for -- either
                # way the event below marks its end, the calling (encoder) stream goes on, and in the first case the exchange
                # sits on the hardware queue functional.assign_streams chose for it (beside a weight-gradient companion)
                # instead of wherever the runtime puts the group's own stream.
                cur = torch.cuda.current_stream()
                comm = aux_stream(self.flat_grad.device, "comm")
                ws = wgrad_stream_of(cur)
                if ws is not None:
                    comm.wait_stream(ws)    # the tail's weight gradients are forked onto the companion stream: the exchange
                comm.wait_stream(cur)       # waits for them, the backward chain on `cur` does not
                with torch.cuda.stream(comm):
                    dist.all_reduce(self.flat_grad[self.tail_lo:], op=dist.ReduceOp.SUM)
                    ev = torch.cuda.Event()
                    ev.record(comm)
                self._tail_work = ev
                return
            self._tail_work = dist.all_reduce(self.flat_grad[self.tail_lo:], op=dist.ReduceOp.SUM, async_op=True)

    def broadcast_parameters(self, extra=()):
        if self.world > 1:
            dist.broadcast(self.flat_param, src=0)
            for t in extra:
                dist.broadcast(t, src=0)

    def measure_exposed(self, on=True):
        """record a HIP event pair around every all_reduce_grads() call (the part of the exchange the step waits for:
        everything behind the end of backward); exposed_ms() reads them"""
        self._timing = ([], []) if on and self.flat_grad.is_cuda else None

    def exposed_ms(self):
        """mean milliseconds per step between the end of backward and the end of the gradient exchange"""
        if not self._timing or not self._timing[0]:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in zip(*self._timing)]
        return sum(ms) / len(ms)

    def describe(self):
        """what the process group really is (bench.py prints it: a line from an N-GPU run must prove RCCL saw N ranks)"""
        if not dist.is_initialized():
            return {"backend": None, "world_size": 1}
        out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank": dist.get_rank(),
               "dp_overlap": self.tail_lo is not None,
               "buckets_bytes": ([4 * self.tail_lo, 4 * (self.flat_grad.numel() - self.tail_lo)] if self.tail_lo is not None
                                 else [4 * self.flat_grad.numel()])}
        # one collective whose result only the right world size gives: sum over ranks of (rank + 1)
        t = torch.tensor([float(dist.get_rank() + 1)], dtype=torch.float64, device=self.flat_grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out["rank_sum"] = float(t.item())
        out["rank_sum_expected"] = self.world * (self.world + 1) / 2.0
        return out

    def all_reduce_grads(self):
        if self.world > 1:
            timed = self._timing is not None and len(self._timing[0]) < 4096      # (a bounded sample: a long run must not grow the lists)
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self._timing[0].append(e)
            self._all_reduce_grads()
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self._timing[1].append(e)

    def _all_reduce_grads(self):
        if self.world > 1:
            if self.flat_grad.is_cuda:
                from .functional import join_aux_streams
                join_aux_streams()      # weight gradients are produced on auxiliary HIP streams
            if self._tail_work is not None:
                dist.all_reduce(self.flat_grad[:self.tail_lo], op=dist.ReduceOp.SUM)
                if isinstance(self._tail_work, torch.cuda.Event):
                    torch.cuda.current_stream().wait_event(self._tail_work)
                else:
                    self._tail_work.wait()
                self._tail_work = None
            else:
                dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)

    def max_over_ranks(self, value):
        """max of a python float over ranks (bench timing)"""
        if self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.flat_grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
