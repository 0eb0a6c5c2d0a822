"""ORACLE (test infrastructure only, see oracle/__init__.py): numpy restatement of the
reference's scan projection.  Follows deeplio/common/laserscan.py:122-185 (do_range_projection),
:215-248 (do_normal_projection) and deeplio/datasets/kitti.py:83-97 + :345-364 (get_velo_image +
transform_images, normalised branch) line by line; pinned bit-exactly (indices) by
tests/golden/projection.npz, which was captured from the reference itself."""
import numpy as np


def range_projection(points, remissions, H, W, fov_up, fov_down, exact_trig=False):
    """exact_trig=False: numpy's own float32 arctan2 / arcsin, as the reference calls them (their
    last-ulp results depend on the host's SIMD dispatch); exact_trig=True: correctly rounded float32
    values (fp64 evaluation rounded once), the host-independent definition the HIP kernel implements
    (csrc/projection.hip).  The two differ for ~1e-5 of the points of a KITTI-sized cloud;
    tests/golden/projection_kitti.npz records that set for this container's numpy."""
    fov_up = fov_up / 180.0 * np.pi
    fov_down = fov_down / 180.0 * np.pi
    fov = abs(fov_down) + abs(fov_up)
    depth = np.linalg.norm(points, 2, axis=1)
    scan_x, scan_y, scan_z = points[:, 0], points[:, 1], points[:, 2]
    if exact_trig:
        yaw = -np.arctan2(scan_y.astype(np.float64), scan_x.astype(np.float64)).astype(np.float32)
        pitch = np.arcsin((scan_z / depth).astype(np.float64)).astype(np.float32)
    else:
        yaw = -np.arctan2(scan_y, scan_x)
        pitch = np.arcsin(scan_z / depth)
    proj_x = 0.5 * (yaw / np.pi + 1.0)
    proj_y = 1.0 - (pitch + abs(fov_down)) / fov
    proj_x *= W
    proj_y *= H
    proj_x = np.floor(proj_x)
    proj_x = np.minimum(W - 1, proj_x)
    proj_x = np.maximum(0, proj_x).astype(np.int32)
    proj_y = np.floor(proj_y)
    proj_y = np.minimum(H - 1, proj_y)
    proj_y = np.maximum(0, proj_y).astype(np.int32)
    out = dict(proj_x=proj_x.copy(), proj_y=proj_y.copy(), unproj_range=depth.copy())
    indices = np.arange(depth.shape[0])
    # the reference uses np.argsort(depth)[::-1] (ties undefined); a stable sort on (-depth, -index)
    # makes the smaller index win among equal depths, the rule the HIP kernel documents
    order = np.lexsort((-indices, -depth.astype(np.float64)))
    proj_range = np.zeros((H, W), np.float32)
    proj_xyz = np.zeros((H, W, 3), np.float32)
    proj_remission = np.zeros((H, W), np.float32)
    proj_idx = np.zeros((H, W), np.int32)
    py, px = proj_y[order], proj_x[order]
    proj_range[py, px] = depth[order]
    proj_xyz[py, px] = points[order]
    proj_remission[py, px] = remissions[order]
    proj_idx[py, px] = indices[order]
    out.update(proj_range=proj_range, proj_xyz=proj_xyz, proj_remission=proj_remission, proj_idx=proj_idx,
               proj_mask=(proj_idx > 0).astype(np.int32))
    return out


def normal_projection(proj_xyz, proj_range):
    img = np.dstack((proj_xyz, proj_range))

    def calc_weights(x, alpha=-0.8):
        return np.exp(alpha * np.abs(x))

    dv = img[:-1, :, :] - img[1:, :, :]
    dh = img[:, :-1, :] - img[:, 1:, :]
    top, bottom = dv[:-1, 1:-1, :], -dv[1:, 1:-1, :]
    left, right = dh[1:-1, :-1, :], -dh[1:-1, 1:, :]
    w = calc_weights(np.stack((top[:, :, -1], left[:, :, -1], bottom[:, :, -1], right[:, :, -1]), axis=2))
    n_tl = np.cross(w[..., 0:1] * top[..., :3], w[..., 1:2] * left[..., :3])
    n_lb = np.cross(w[..., 1, None] * left[..., :3], w[..., 2, None] * bottom[..., :3])
    n_br = np.cross(w[..., 2, None] * bottom[..., :3], w[..., 3, None] * right[..., :3])
    n_rt = np.cross(w[..., 3, None] * right[..., :3], w[..., 0, None] * top[..., :3])
    n = np.sum(np.stack((n_tl, n_lb, n_br, n_rt)), axis=0)
    n /= (np.linalg.norm(n, axis=2, keepdims=True) + 1e-8)
    return np.pad(n, ((1, 1), (1, 1), (0, 0)))


def velo_image(proj_xyz, proj_remission, normals, proj_range, max_depth, channels, mean=None, crop_top=0,
               crop_left=0):
    image = np.dstack((proj_xyz / max_depth, proj_remission, normals, proj_range))
    ct, cl = crop_top, crop_left
    H, W = image.shape[:2]
    img = image[ct:H - ct, cl:W - cl, :].transpose(2, 0, 1).astype(np.float32).copy()
    if mean is not None:
        img -= np.asarray(mean, np.float32)[:, None, None]
    return img[list(channels)]
