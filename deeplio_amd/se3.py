"""Trainer.se3_to_SE3 (trainer.py:324-351) / Tester.se3_to_SE3 (tester.py:223-251) as one
batched HIP launch with a hand-written backward."""
import torch

from . import functional as Fh


def se3_to_SE3(f2f_x, f2f_r, ordering="wxyz", status=None):
    """f2f_x, f2f_r: [B,S,3] -> (f2g_x [B,S,3], f2g_q [B,S,4]).  `status` (int32[1] on the
    device) accumulates bit 0 = determinant check failed (the reference raises ValueError),
    bit 1 = chained rotation left SO(3) by more than liegroups' 1e-6 tolerance (the reference
    would re-orthonormalise through an SVD).  Read it with check_status() when convenient --
    no per-step host sync is forced."""
    return Fh.SE3ChainFn.apply(f2f_x, f2f_r, 0 if ordering == "wxyz" else 1, status)


def check_status(status):
    v = int(status.item())
    if v & 1:
        raise ValueError("Det error: a chained rotation has det != 1")
    return v
