"""DataCombiCreater (deeplio/models/misc.py:24-125) restated on CPU: pair gather + channel
split (misc.py:65-69) and the ground-truth transform (misc.py:83-125: inv_SE3 of
common/spatial.py:904-923, relative 4x4 products, SO3.log / to_quaternion of the restated
liegroups in oracle/se3.py).  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import numpy as np
import torch

from . import se3


def process_images(imgs, combinations):
    g = imgs[:, np.asarray(combinations)]          # [B, S, T, C, H, W]
    return g[:, :, :, 0:3], g[:, :, :, 3:].contiguous()


def inv_SE3(T):
    Tn = T.numpy()
    R, t = Tn[:3, :3], Tn[:3, 3]
    Ti = np.eye(4)
    Ti[:3, :3] = R.T
    Ti[:3, 3] = -np.matmul(R.T, t)
    return torch.from_numpy(Ti).type(T.dtype)


def process_ground_truth(gts, combinations):
    """gts [S+1, 15] rows [x(3), R(9), v(3)] -> (f2f [S,6], f2g [S,7])"""
    Ts = []
    for g in gts:
        T = torch.eye(4)
        T[:3, 3] = g[0:3]
        T[:3, :3] = g[3:12].reshape(3, 3)
        Ts.append(T)
    f2f, f2g = [], []
    T0_inv = inv_SE3(Ts[0])
    for a, b in combinations:
        rel = torch.matmul(inv_SE3(Ts[a]), Ts[b])
        f2f.append(torch.cat([rel[:3, 3], se3.so3_log(rel[:3, :3])]))
        glo = torch.matmul(T0_inv, Ts[b])
        f2g.append(torch.cat([glo[:3, 3], se3.rot_to_quat(glo[:3, :3])]))
    return torch.stack(f2f), torch.stack(f2g)
