import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplio_amd.laserscan import LaserScan
G = np.load("tests/golden/projection.npz")
s = LaserScan(H=64, W=512, device="cuda"); s.set_points(torch.from_numpy(G["points"]).cuda(), torch.from_numpy(G["remissions"]).cuda()); s.do_range_projection()
P = G["points"]; d_np = np.linalg.norm(P, 2, axis=1); d_hip = s.unproj_range.cpu().numpy()
bad = np.nonzero(d_np != d_hip)[0]
print("mismatching depths", len(bad), "of", len(d_np))
x, y, z = (P[:, i] for i in range(3))
alts = {"(xx+yy)+zz": np.sqrt((x*x+y*y)+z*z), "xx+(yy+zz)": np.sqrt(x*x+(y*y+z*z)),
        "f64": np.sqrt(x.astype(np.float64)**2+y.astype(np.float64)**2+z.astype(np.float64)**2).astype(np.float32),
        "fma(z,z,fma(y,y,xx))": None}
for k, v in alts.items():
    if v is not None: print(k, "eq numpy:", int((v != d_np).sum()), " eq hip:", int((v != d_hip).sum()))
for i in bad[:5]: print(P[i], d_np[i].view(np.int32), d_hip[i].view(np.int32))
