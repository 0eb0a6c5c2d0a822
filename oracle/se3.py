"""SO(3)/SE(3) helpers -- restatement of liegroups.torch.SO3 (utiasSTARS/liegroups, PyPI
`liegroups`; the reference does not pin a version and does not vendor it) as used at
  trainer.py:339 (SO3.exp(w).as_matrix()), trainer.py:349 (from_matrix(normalize=True)
  .to_quaternion()), misc.py:104 (from_matrix(normalize=False).log()), misc.py:119
  (from_matrix().to_quaternion()),
and of Trainer.se3_to_SE3 (trainer.py:324-351) / Tester.se3_to_SE3 (tester.py:223-251).
Pure torch, differentiable (autograd provides the reference gradients).
TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch

TOL = 1e-6  # liegroups.torch.utils.isclose tolerance


def wedge(phi):
    z = torch.zeros((), dtype=phi.dtype)
    return torch.stack([torch.stack([z, -phi[2], phi[1]]),
                        torch.stack([phi[2], z, -phi[0]]),
                        torch.stack([-phi[1], phi[0], z])])


def so3_exp(phi):
    """Rodrigues; first-order I + phi^ when |phi| < 1e-6 (liegroups SO3.exp)."""
    angle = phi.norm(p=2)
    eye = torch.eye(3, dtype=phi.dtype)
    if angle.abs() < TOL:
        return eye + wedge(phi)
    axis = phi / angle
    s, c = torch.sin(angle), torch.cos(angle)
    return c * eye + (1. - c) * torch.outer(axis, axis) + s * wedge(axis)


def so3_log(R):
    """liegroups SO3.log: acos of the clamped cosine; small-angle vee(R - I)."""
    cos_angle = (0.5 * torch.trace(R) - 0.5).clamp(-1., 1.)
    angle = torch.acos(cos_angle)
    if angle.abs() < TOL:
        M = R - torch.eye(3, dtype=R.dtype)
    else:
        M = (0.5 * angle / torch.sin(angle)) * (R - R.t())
    return torch.stack([M[2, 1], M[0, 2], M[1, 0]])


def is_valid_rotation(R):
    ok_det = (torch.det(R) - 1.).abs() < TOL
    ok_orth = ((R.t() @ R - torch.eye(3, dtype=R.dtype)).abs() < TOL).all()
    return bool(ok_det and ok_orth)


def normalize_rotation(R):
    """liegroups SO3.normalize: project onto SO(3) through the SVD."""
    U, _, Vh = torch.linalg.svd(R)
    S = torch.eye(3, dtype=R.dtype)
    S[2, 2] = torch.det(U) * torch.det(Vh)
    return U @ S @ Vh


def rot_to_quat(R, ordering="wxyz"):
    """liegroups SO3.to_quaternion."""
    qw = 0.5 * torch.sqrt(1. + R[0, 0] + R[1, 1] + R[2, 2])
    if not (qw.abs() < TOL):
        d = 4. * qw
        qx, qy, qz = (R[2, 1] - R[1, 2]) / d, (R[0, 2] - R[2, 0]) / d, (R[1, 0] - R[0, 1]) / d
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        d = 2. * torch.sqrt(1. + R[0, 0] - R[1, 1] - R[2, 2])
        qw, qx, qy, qz = (R[2, 1] - R[1, 2]) / d, 0.25 * d, (R[1, 0] + R[0, 1]) / d, (R[0, 2] + R[2, 0]) / d
    elif R[1, 1] > R[2, 2]:
        d = 2. * torch.sqrt(1. + R[1, 1] - R[0, 0] - R[2, 2])
        qw, qx, qy, qz = (R[0, 2] - R[2, 0]) / d, (R[1, 0] + R[0, 1]) / d, 0.25 * d, (R[2, 1] + R[1, 2]) / d
    else:
        d = 2. * torch.sqrt(1. + R[2, 2] - R[0, 0] - R[1, 1])
        qw, qx, qy, qz = (R[1, 0] - R[0, 1]) / d, (R[0, 2] + R[2, 0]) / d, (R[2, 1] + R[1, 2]) / d, 0.25 * d
    if ordering == "wxyz":
        return torch.stack([qw, qx, qy, qz])
    return torch.stack([qx, qy, qz, qw])


def se3_to_SE3(f2f_x, f2f_r, ordering="wxyz", check=True):
    """Trainer.se3_to_SE3 (trainer.py:324-351).  ordering='xyzw' gives the tester.py:223-251
    variant.  Raises ValueError on the reference's determinant checks."""
    B, S, _ = f2f_x.shape
    qs, xs = [], []
    for b in range(B):
        R_prev = torch.eye(3, dtype=f2f_x.dtype)
        t_prev = torch.zeros(3, dtype=f2f_x.dtype)
        qb, xb = [], []
        for s in range(S):
            R_cur = so3_exp(f2f_r[b, s])
            if check and not torch.isclose(torch.det(R_cur), torch.ones((), dtype=R_cur.dtype)):
                raise ValueError("Det error:\nR\n{}\nq:\n{}".format(R_cur, f2f_r[b, s]))
            t_prev = R_prev @ f2f_x[b, s] + t_prev
            R_prev = R_prev @ R_cur
            if check and not torch.isclose(torch.det(R_prev), torch.ones((), dtype=R_prev.dtype)):
                raise ValueError("Det error:\nR\n{}".format(R_prev))
            Rn = R_prev if is_valid_rotation(R_prev.detach()) else normalize_rotation(R_prev)
            qb.append(rot_to_quat(Rn, ordering))
            xb.append(t_prev)
        qs.append(torch.stack(qb))
        xs.append(torch.stack(xb))
    return torch.stack(xs), torch.stack(qs)
