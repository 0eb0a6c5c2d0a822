"""The serial middle of the training step as ONE hipGraph launch.

Between the last encoder layer and the first encoder-backward kernel the step is a chain of ~200 small
dependent launches: lidar feature fusion + fc1, fusion net, odometry net (per-timestep RNN cells), the two
heads, NaN/Inf flags, the SE(3) chain, the loss (Trainer.train, trainer.py:213-281 from `self.model(...)`'s
last layers to `loss.backward()`'s first ones) and the backward of all of that.  Each runs 2-30 us on the
GPU but costs ~15 us of host time to issue and nothing else can run beside them (every stream waits for the
loss), so the GPU idles ~3-4 ms per step (gpurun r02 step dump: 10.3 -> 14.6 ms of a 28.8 ms step).

TailGraph records that chain -- forward, loss AND backward down to the encoder / IMU features -- once with
torch.cuda.CUDAGraph (= hipGraph on ROCm) and replays it per step:

* inputs (encoder features a/b, IMU feature, ground truth) are copied into static buffers;
* parameter gradients are accumulated by the recorded kernels straight into the optimizer's flat gradient
  buffer (functional._sink), parameters are read from the flat parameter buffer: both have fixed addresses;
* dropout masks come from Philox offsets read on the device (dlio_dropout_fwd_at): a replay draws the masks the
  eager step would have drawn;
* outputs are the loss and the gradients w.r.t. the three features; the caller (TrainStep) feeds those into
  the IMU net's and the encoders' own tapes, which stay eager and multi-stream.

Same kernels, same order, same arguments as the eager step: the results are bit-identical
(tests/test_gpu_model.py::test_tail_graph_matches_eager).
"""
import torch

from . import functional as Fh


def _sig(t):
    return None if t is None else (tuple(t.shape), t.dtype, bool(t.requires_grad))


class TailGraph:
    def __init__(self, step):
        self.step = step            # TrainStep: model, criterion, flags, _tail()
        self.graph = None
        self.signature = None

    @staticmethod
    def signature_of(feats, gts_f2f, gts_f2g):
        enc = feats["lidar"]
        return (None if enc is None else (_sig(enc[0]), _sig(enc[1]), tuple(enc[2])), _sig(feats["imu"]),
                tuple(gts_f2f.shape), tuple(gts_f2g.shape))

    def capture(self, feats, gts_f2f, gts_f2g):
        """record forward + loss + backward of the tail for features of these shapes"""
        def leaf(t):
            s = t.detach().clone()
            s.requires_grad_(t.requires_grad)
            return s

        dev = gts_f2f.device
        enc = feats["lidar"]
        self.signature = self.signature_of(feats, gts_f2f, gts_f2g)
        self.fa = self.fb = self.fi = None
        if enc is not None:
            self.fa, self.fb, self.bs = leaf(enc[0]), leaf(enc[1]), tuple(enc[2])
        if feats["imu"] is not None:
            self.fi = leaf(feats["imu"])
        self.gt_f2f, self.gt_f2g = gts_f2f.detach().clone(), gts_f2g.detach().clone()
        self.base = torch.zeros(1, dtype=torch.int64, device=dev)
        static = {"lidar": None if enc is None else (self.fa, self.fb, self.bs), "imu": self.fi, "imu_stream": None}
        for p in self.step.optimizer.params:        # the recorded kernels accumulate into these views
            if p.requires_grad and p.grad is None:
                raise RuntimeError("TailGraph.capture needs the flat gradient views (optimizer.zero_grad() first)")
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        Fh._NO_JOIN[0] = True       # a capturing stream must not wait for work outside the capture
        try:
            with torch.cuda.graph(graph):
                with Fh.dropout_base(self.base) as db:
                    loss = self.step._tail(static, self.gt_f2f, self.gt_f2g, hook=False)
                loss.backward()
        finally:
            Fh._NO_JOIN[0] = False
        self.consumed = db.consumed
        self.loss = loss.detach()
        self.grads = tuple(None if t is None or not t.requires_grad else t.grad for t in (self.fa, self.fb, self.fi))
        self.graph = graph
        return self

    def replay(self, feats, gts_f2f, gts_f2g):
        """-> (loss, (d feature a, d feature b, d IMU feature)); static tensors, valid until the next replay"""
        enc = feats["lidar"]
        with torch.no_grad():
            if enc is not None:
                self.fa.copy_(enc[0])
                self.fb.copy_(enc[1])
            if self.fi is not None:
                self.fi.copy_(feats["imu"])
            self.gt_f2f.copy_(gts_f2f)
            self.gt_f2g.copy_(gts_f2g)
            if self.consumed:
                self.base.fill_(Fh.dropout_offset())
                Fh.advance_dropout(self.consumed)
        self.graph.replay()
        return self.loss, self.grads
