"""informational: the reference's `Inf-Time` (model forward in test mode, S=1) on the HIP path"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deeplio_amd.config import make_config
from deeplio_amd.tester import TestStep
dev = torch.device("cuda", 0)
for B in (1, 8):
    cfg = make_config(seq=1)
    ts = TestStep(cfg, (5, 64, 2048), dev, batch_size=B)
    batch = bench.synth_batch(1, B, 1, 5, 64, 2048, 50, dev)
    for _ in range(5): ts.step(*batch)
    ts.inference_time = 0.0; ts.steps = 0
    for _ in range(20): ts.step(*batch, timed=True)
    print("PointSeg headline model, test mode, B=%d S=1: Inf-Time %.2f ms per batch (%.1f frame pairs/s)" % (
        B, 1e3 * ts.inference_time / ts.steps, B * ts.steps / ts.inference_time))
    try:
        ts.capture(*batch[:3])
        ref = ts.model([[batch[0], batch[1]], batch[2]])
        out = ts.forward_graph(*batch[:3])
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ts.forward_graph(*batch[:3])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print("   hipGraph replay: %.2f ms per batch (%.1f frame pairs/s), bit-identical outputs" % (1e3 * dt, B / dt))
    except Exception as e:      # noqa
        print("   hipGraph capture failed:", type(e).__name__, str(e)[:300])
