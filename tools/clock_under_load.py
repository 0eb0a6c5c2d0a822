"""what shader clock does the part sustain under (a) the pure-MFMA micro kernel, (b) the 3x3 conv
forward kernel?  Samples rocm-smi while the kernels loop."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeplio_amd import ops
dev = torch.device("cuda:0")
def sample(tag, stop):
    vals = []
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out); p = re.search(r"Power \(W\):\s*([\d.]+)", out) or re.search(r"Socket Power \(W\):\s*([\d.]+)", out)
            if m: vals.append((int(m.group(1)), float(p.group(1)) if p else -1))
        except Exception as e:
            vals.append((-1, -1))
        time.sleep(0.2)
    print(tag, "samples (MHz, W):", vals[:12])
N, ci, co, H, W = 16, 64, 256, 64, 128
x = torch.randn(N, ci, H, W, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
y = torch.empty(N, co, H, W, device=dev)
d = ops.conv_desc(N, ci, H, W, co, 3, 3, 1, 1, 1, 1)
wt = ops.conv2d_prep_weight(w, 0)
stop = threading.Event(); th = threading.Thread(target=sample, args=("conv3x3 64->256", stop)); th.start()
t0 = time.time()
while time.time() - t0 < 3.0:
    for _ in range(200): ops.conv2d_fwd(x, wt, None, y, d)
    torch.cuda.synchronize()
stop.set(); th.join()
stop = threading.Event(); th = threading.Thread(target=sample, args=("idle", stop)); th.start(); time.sleep(1.5); stop.set(); th.join()
