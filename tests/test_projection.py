"""Scan projection.  CPU part: the oracle against the golden captured from the
reference.  GPU part (-m gpu): the HIP kernels through the C-ABI against the golden (bit-exact
int32 indices), against the oracle on a KITTI-sized cloud, and the size-independent
closest-point-wins invariant."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "projection.npz"))
FOV = dict(fov_up=3.0, fov_down=-25.0)


def synth_cloud(seed, n):
    rng = np.random.default_rng(seed)
    az = rng.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(rng.uniform(-24.5, 2.5, n))
    r = rng.uniform(2.0, 60.0, n)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)
    return pts, rng.uniform(0, 1, n).astype(np.float32)


def test_oracle_matches_reference_golden():
    from oracle import projection as op
    o = op.range_projection(GOLD["points"], GOLD["remissions"], 64, 512, **FOV)
    for k in ("proj_x", "proj_y", "proj_idx"):                      # bit-exact index work
        assert np.array_equal(o[k], GOLD[k]), k
    for k in ("proj_range", "proj_xyz", "proj_remission"):
        assert np.array_equal(o[k], GOLD[k]), k
    n = op.normal_projection(o["proj_xyz"], o["proj_range"])
    assert np.array_equal(n.astype(np.float32), GOLD["normals"])


KITTI = np.load(os.path.join(HERE, "golden", "projection_kitti.npz"))


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_kitti_maps(px, py, idx, rng_img):
    """px, py, idx computed with correctly rounded arctan2 / arcsin (the host-independent definition:
    oracle exact_trig=True, the HIP kernel) against the reference's maps at the headline geometry
    (64x2048, 120 k points; tests/golden/projection_kitti.npz, captured from LaserScan.do_range_projection):
    the recorded points / pixels -- where numpy's float32 trig on the capturing host lands in the
    neighbouring pixel -- hold the recorded 'exact' values, and with the recorded reference values
    patched in the maps reproduce the reference's SHA-256 digests, i.e. they are BIT-EXACT everywhere
    else."""
    g = KITTI
    mp, mx = g["mis_point"], g["mis_pixel"]
    assert len(mp) <= 4 and len(mx) <= 8            # what the fixture records: 1 point, 2 pixels
    assert np.array_equal(px[mp], g["mis_exact_x"]) and np.array_equal(py[mp], g["mis_exact_y"])
    assert np.array_equal(idx.reshape(-1)[mx], g["mis_pixel_exact_idx"])
    px, py, idx, rng_img = px.copy(), py.copy(), idx.copy().reshape(-1), rng_img.copy().reshape(-1)
    px[mp], py[mp] = g["mis_ref_x"], g["mis_ref_y"]
    idx[mx] = g["mis_pixel_ref_idx"]
    rng_img[mx] = g["mis_pixel_ref_range"]
    assert _sha(px.astype(np.int32)) == str(g["ref_proj_x_sha256"])
    assert _sha(py.astype(np.int32)) == str(g["ref_proj_y_sha256"])
    assert _sha(idx.astype(np.int32)) == str(g["ref_proj_idx_sha256"])
    assert _sha(rng_img.astype(np.float32)) == str(g["ref_proj_range_sha256"])
    assert int((idx > 0).sum()) == int(g["n_occupied"])


def test_oracle_kitti_size_maps_vs_reference_golden():
    from oracle import projection as op
    pts, rem = synth_cloud(int(KITTI["seed"]), int(KITTI["n"]))
    assert _sha(pts) == str(KITTI["points_sha256"])
    H, W = int(KITTI["H"]), int(KITTI["W"])
    ex = op.range_projection(pts, rem, H, W, exact_trig=True, **FOV)
    _check_kitti_maps(ex["proj_x"], ex["proj_y"], ex["proj_idx"], ex["proj_range"])
    # numpy's own float32 trig (the reference's call): identical to the golden on the capturing host; on
    # another host the differing set may be another handful of points -- then only report
    o = op.range_projection(pts, rem, H, W, **FOV)
    if _sha(o["proj_x"].astype(np.int32)) == str(KITTI["ref_proj_x_sha256"]):
        assert _sha(o["proj_idx"].astype(np.int32)) == str(KITTI["ref_proj_idx_sha256"])
    else:
        d = np.nonzero((o["proj_x"] != ex["proj_x"]) | (o["proj_y"] != ex["proj_y"]))[0]
        assert len(d) <= 12


def test_oracle_velo_image_layout():
    from oracle import projection as op
    o = op.range_projection(GOLD["points"], GOLD["remissions"], 64, 512, **FOV)
    n = op.normal_projection(o["proj_xyz"], o["proj_range"]).astype(np.float32)
    mean = np.arange(8, dtype=np.float32) * 0.1
    img = op.velo_image(o["proj_xyz"], o["proj_remission"], n, o["proj_range"], 80, [0, 1, 2, 4, 7], mean, 2, 4)
    assert img.shape == (5, 60, 504)
    assert np.allclose(img[4], o["proj_range"][2:-2, 4:-4] - 0.7)
    assert np.allclose(img[0], o["proj_xyz"][2:-2, 4:-4, 0] / 80)


# ------------------------------------------------------------------------------------ GPU
def _hip_scan(dev, pts, rem, H, W):
    from deeplio_amd.laserscan import LaserScan
    s = LaserScan(project=False, H=H, W=W, device=dev, **FOV)
    s.set_points(torch.from_numpy(pts).to(dev), torch.from_numpy(rem).to(dev))
    s.do_range_projection()
    s.do_normal_projection()
    return s


@pytest.mark.gpu
def test_hip_projection_vs_reference_golden(dev):
    s = _hip_scan(dev, GOLD["points"], GOLD["remissions"], 64, 512)
    for k in ("proj_x", "proj_y", "proj_idx"):                      # bit-exact
        assert np.array_equal(getattr(s, k).cpu().numpy(), GOLD[k]), k
    for k in ("proj_range", "proj_xyz", "proj_remission"):          # gathers of exact values
        assert np.array_equal(getattr(s, k).cpu().numpy(), GOLD[k]), k
    assert np.array_equal(s.proj_mask.cpu().numpy(), (GOLD["proj_idx"] > 0).astype(np.int32))
    # normals: float path (exp, cross products, normalisation): 1e-4 absolute on unit vectors,
    # except where the un-normalised sum is itself rounding noise (flat, collinear neighbourhoods)
    n, g = s.proj_normal.cpu().numpy(), GOLD["normals"]
    err = np.abs(n - g).max(axis=2)
    assert np.mean(err < 1e-4) > 0.999
    assert np.all((err < 1e-4) | (np.abs(np.linalg.norm(g, axis=2) - 1) > 1e-3) | _ill_conditioned(GOLD, err))


def _ill_conditioned(gold, err):
    """pixels whose normal is the normalisation of a cancelling sum: |sum| << sum of |terms|"""
    from oracle import projection as op
    img = np.dstack((gold["proj_xyz"], gold["proj_range"])).astype(np.float64)
    dv = img[:-1] - img[1:]
    dh = img[:, :-1] - img[:, 1:]
    top, bottom, left, right = dv[:-1, 1:-1], -dv[1:, 1:-1], dh[1:-1, :-1], -dh[1:-1, 1:]
    w = np.exp(-0.8 * np.abs(np.stack((top[..., 3], left[..., 3], bottom[..., 3], right[..., 3]), 2)))
    t = [np.cross(w[..., a, None] * p[..., :3], w[..., b, None] * q[..., :3])
         for (a, p), (b, q) in zip([(0, top), (1, left), (2, bottom), (3, right)],
                                   [(1, left), (2, bottom), (3, right), (0, top)])]
    tot = np.linalg.norm(sum(t), axis=2)
    mag = sum(np.linalg.norm(x, axis=2) for x in t) + 1e-30
    return np.pad(tot / mag < 1e-2, ((1, 1), (1, 1)), constant_values=True)


@pytest.mark.gpu
def test_hip_projection_kitti_size_bit_exact_vs_reference_golden(dev):
    """north_star: projection index maps bit-exact -- at the headline geometry (64x2048, 120 k points)
    against the maps the reference itself produced (see _check_kitti_maps), and identical to the
    oracle's host-independent (correctly rounded trig) evaluation in every element"""
    from oracle import projection as op
    pts, rem = synth_cloud(int(KITTI["seed"]), int(KITTI["n"]))
    H, W = int(KITTI["H"]), int(KITTI["W"])
    s = _hip_scan(dev, pts, rem, H, W)
    px, py = s.proj_x.cpu().numpy(), s.proj_y.cpu().numpy()
    idx, rng_img = s.proj_idx.cpu().numpy(), s.proj_range.cpu().numpy()
    _check_kitti_maps(px, py, idx, rng_img)
    ex = op.range_projection(pts, rem, H, W, exact_trig=True, **FOV)
    for k, v in (("proj_x", px), ("proj_y", py), ("proj_idx", idx), ("proj_range", rng_img),
                 ("proj_xyz", s.proj_xyz.cpu().numpy()), ("proj_remission", s.proj_remission.cpu().numpy()),
                 ("unproj_range", s.unproj_range.cpu().numpy())):
        assert np.array_equal(v, ex[k]), k


@pytest.mark.gpu
def test_hip_projection_kitti_size_vs_oracle(dev):
    """120k points into 64x2048.  numpy's float32 arctan2/arcsin on this host (SVML) are up to a few
    ulp off the correctly rounded value the kernel computes, so an index may differ where
    0.5*(yaw/pi+1)*W sits within ~1e-4 of an integer: allow <= 1e-4 of the points, and require
    every differing point to be such a boundary case."""
    from oracle import projection as op
    pts, rem = synth_cloud(77, 120000)
    H, W = 64, 2048
    o = op.range_projection(pts, rem, H, W, **FOV)
    s = _hip_scan(dev, pts, rem, H, W)
    px, py = s.proj_x.cpu().numpy(), s.proj_y.cpu().numpy()
    assert np.array_equal(s.unproj_range.cpu().numpy(), o["unproj_range"])      # sqrt chain is exact
    bad = np.nonzero((px != o["proj_x"]) | (py != o["proj_y"]))[0]
    assert len(bad) <= 12, len(bad)
    x, y, z = (pts[bad, i].astype(np.float64) for i in range(3))
    vx = 0.5 * (-np.arctan2(y, x) / np.pi + 1.0) * W
    fd, fu = 25.0 / 180 * np.pi, 3.0 / 180 * np.pi
    vy = (1.0 - (np.arcsin(z / np.sqrt(x * x + y * y + z * z)) + fd) / (fd + fu)) * H
    near = (np.abs(vx - np.round(vx)) < 2e-3) | (np.abs(vy - np.round(vy)) < 2e-3)
    assert np.all(near)
    assert np.all(np.abs(px[bad] - o["proj_x"][bad]) <= 1) and np.all(np.abs(py[bad] - o["proj_y"][bad]) <= 1)
    if len(bad) == 0:
        for k in ("proj_idx", "proj_range", "proj_xyz", "proj_remission"):
            assert np.array_equal(getattr(s, k).cpu().numpy(), o[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W", [(120000, 64, 2048), (500, 16, 64), (1, 8, 8), (0, 8, 8)])
def test_hip_projection_closest_wins_invariant(dev, n, H, W):
    """size-independent property: every occupied pixel holds the closest of the points that fall
    into it (smallest index among equal depths); every point's pixel is occupied; empty pixels are 0."""
    pts, rem = synth_cloud(5, max(n, 1))
    pts, rem = pts[:n], rem[:n]
    if n > 10:                       # exact duplicates -> depth ties
        pts[7], pts[3] = pts[9], pts[9]
    s = _hip_scan(dev, pts, rem, H, W)
    px, py = s.proj_x.cpu().numpy().astype(np.int64), s.proj_y.cpu().numpy().astype(np.int64)
    depth = s.unproj_range.cpu().numpy()
    idx = s.proj_idx.cpu().numpy()
    rng_img = s.proj_range.cpu().numpy()
    pix = py * W + px
    best = {}
    for i in range(n):
        b = best.get(pix[i])
        if b is None or depth[i] < depth[b]:
            best[pix[i]] = i
    occ = np.zeros(H * W, bool)
    for p, i in best.items():
        occ[p] = True
        assert idx.reshape(-1)[p] == i and rng_img.reshape(-1)[p] == depth[i]
        assert np.array_equal(s.proj_xyz.cpu().numpy().reshape(-1, 3)[p], pts[i])
    assert np.all(idx.reshape(-1)[~occ] == 0) and np.all(rng_img.reshape(-1)[~occ] == 0)
    assert float(s.proj_xyz.cpu().numpy().reshape(-1, 3)[~occ].__abs__().sum()) == 0.0


@pytest.mark.gpu
def test_hip_velo_image_vs_oracle(dev):
    from oracle import projection as op
    s = _hip_scan(dev, GOLD["points"], GOLD["remissions"], 64, 512)
    mean = [0.01, -0.02, 0.03, 0.2, 0.0, 0.1, -0.1, 12.0]
    for channels, ct, cl in (([0, 1, 2, 3, 4, 5, 6, 7], 0, 0), ([0, 1, 2, 4, 7], 2, 4), ([7, 3], 0, 8)):
        got = s.velo_image(channels, mean, ct, cl).cpu().numpy()
        want = op.velo_image(s.proj_xyz.cpu().numpy(), s.proj_remission.cpu().numpy(),
                             s.proj_normal.cpu().numpy(), s.proj_range.cpu().numpy(), 80, channels, mean, ct, cl)
        assert got.shape == want.shape
        assert np.array_equal(got, want)


def test_projection_rejects_bad_arguments():
    from deeplio_amd import _lib
    lib = _lib.lib
    assert lib.dlio_scan_project_ws_bytes(64, 2048) == 64 * 2048 * 8
    assert lib.dlio_scan_project_ws_bytes(0, 5) == 0
    assert lib.dlio_scan_normals(None, None, None, 4, 4, None) == _lib.DLIO_EINVAL
