"""bf16 vs fp32 HIP path: per-tensor cosine / relative L2 of gradients (diagnosis of the mixed path)"""
import os, sys, types
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import golden_common as gc
from deeplio_amd import nets, mixed
dev = torch.device("cuda:0")


def cmp(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(F.cosine_similarity(a, b, dim=0)), float((a - b).norm() / b.norm().clamp_min(1e-30))


def fire(case):
    N, cin, sq, e, H, W, byp = case
    f16, f32 = nets.Fire(cin, sq, e, e, bypass=byp), nets.Fire(cin, sq, e, e, bypass=byp)
    gc.fill_state(f32, 77); f16.load_state_dict(f32.state_dict())
    f16.to(dev).train(); f32.to(dev).train()
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1)).bfloat16().float()
    g = torch.randn(N, 2 * e, H, W, generator=torch.Generator().manual_seed(2)).bfloat16().float()
    xa = x.to(dev).bfloat16().requires_grad_(True); xb = x.to(dev).requires_grad_(True)
    ya, yb = f16(xa), f32(xb)
    ya.backward(g.to(dev).bfloat16()); yb.backward(g.to(dev))
    print(case, "y", cmp(ya.float(), yb), "dx", cmp(xa.grad.float(), xb.grad))
    for (k, p), (_, q) in zip(f16.named_parameters(), f32.named_parameters()):
        print("   %-22s cos %.4f relL2 %.3e" % ((k,) + cmp(p.grad, q.grad)))


for c in [(2, 64, 16, 64, 8, 32, "simple"), (3, 256, 48, 192, 4, 16, None), (8, 128, 16, 64, 32, 128, "simple")]:
    fire(c)


def model(H, W, B, S):
    from deeplio_amd import losses, misc
    from deeplio_amd.config import make_config
    from deeplio_amd.se3 import se3_to_SE3
    res = {}
    for prec in ("bf16", "fp32"):
        ov = dict(gc.NO_DROP); ov.update(gc.SMALL_RNN); ov['lidar-feat-pointseg/precision'] = prec
        cfg = make_config(seq=S, overrides=ov)
        misc.build_config_container(cfg, types.SimpleNamespace(device="cuda:0", batch_size=B))
        m = nets.get_model((5, H, W), cfg, dev); gc.fill_state(m, 1000); m.train()
        crit = losses.get_loss_function(cfg, dev)
        xyz, nrm, imu, f2f, f2g = (t.to(dev) for t in gc.make_batch(2000, B, S, 5, H, W, 7))
        pt, pw = m([[xyz, nrm], imu]); pp, pq = se3_to_SE3(pt, pw)
        loss = crit(pt, pw, pp[:, 1:3], pq[:, 1:3], f2f[:, :, 0:3], f2f[:, :, 3:], f2g[:, 1:3, 0:3], f2g[:, 1:3, 3:7])
        loss.backward()
        res[prec] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        print(prec, "loss", float(loss))
    for k in res["fp32"]:
        if "encoder1" in k and ("expand3x3.weight" in k or "squeeze.weight" in k or "conv1a.0.weight" in k):
            print("   %-60s cos %.4f relL2 %.3e" % ((k,) + cmp(res["bf16"][k], res["fp32"][k])))
    for k in ("lidar_feat_net.fc1.weight", "odom_feat_net.rnn.weight_ih_l0", "fc_pos.weight"):
        print("   %-60s cos %.4f relL2 %.3e" % ((k,) + cmp(res["bf16"][k], res["fp32"][k])))


model(16, 1024, 2, 2)
model(64, 2048, 1, 2)
