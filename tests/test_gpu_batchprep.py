"""-m gpu: device-side batch preparation (a16, DataCombiCreater) against the oracle and the
reference golden."""
import os
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from conftest import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu


def test_pair_stack_is_bit_exact(dev):
    from deeplio_amd import ops
    from oracle import batchprep
    g = torch.Generator().manual_seed(1)
    imgs = torch.randn(3, 4, 6, 8, 36, generator=g)
    for comb in ([[0, 1], [1, 2], [2, 3]], [[0, 2], [3, 1]]):
        xyz_o, nrm_o = batchprep.process_images(imgs, comb)
        c = torch.tensor(comb, dtype=torch.int32, device=dev)
        xyz, nrm = ops.pair_stack(imgs.to(dev), c, 3)
        assert torch.equal(xyz.cpu(), xyz_o.contiguous()) and torch.equal(nrm.cpu(), nrm_o)


def test_gt_relative_vs_golden_and_oracle(dev):
    from deeplio_amd import misc, ops
    from deeplio_amd.config import make_config
    from oracle import batchprep, se3
    gold = np.load(os.path.join(HERE, "golden", "gt_lr.npz"))
    comb = [[0, 1], [1, 2], [2, 3]]
    c = torch.tensor(comb, dtype=torch.int32, device=dev)
    f2f, f2g = ops.gt_relative(torch.from_numpy(gold['gts'])[None].to(dev), c)
    assert rel_err(f2f[0], torch.from_numpy(gold['f2f'])) < 1e-5      # reference's own numbers
    assert rel_err(f2g[0], torch.from_numpy(gold['f2g'])) < 1e-5
    # batch of random trajectories vs the oracle (includes an identity step: small-angle log branch)
    rng = np.random.default_rng(3)
    B, F = 5, 4
    gts = torch.zeros(B, F, 15)
    for b in range(B):
        R, t = torch.eye(3), torch.zeros(3)
        for f in range(F):
            w = torch.from_numpy((0.3 * rng.standard_normal(3)).astype(np.float32))
            if b == 1 and f == 2:
                w = torch.zeros(3)
            R = R @ se3.so3_exp(w)
            t = t + torch.from_numpy(rng.standard_normal(3).astype(np.float32))
            gts[b, f] = torch.cat([t, R.flatten(), torch.zeros(3)])
    f2f, f2g = ops.gt_relative(gts.to(dev), c)
    for b in range(B):
        o1, o2 = batchprep.process_ground_truth(gts[b], comb)
        assert rel_err(f2f[b], o1) < 1e-5 and rel_err(f2g[b], o2) < 1e-5
    # the mirrored host class
    cfg = make_config(seq=3)
    misc.build_config_container(cfg, types.SimpleNamespace(device=str(dev), batch_size=B))
    dc = misc.DataCombiCreater(np.array(comb), device=dev)
    data = {'images': torch.randn(B, F, 6, 8, 16), 'untrans-images': torch.randn(B, F, 6, 8, 16),
            'imus': torch.rand(B, 3, 5, 6), 'gts': gts}
    dc(data)
    assert dc.res_imgs.shape == (B, 3, 2, 3, 8, 16) and dc.res_normals.shape == (B, 3, 2, 3, 8, 16)
    assert torch.equal(dc.res_gt_f2f, f2f) and torch.equal(dc.res_gt_f2g, f2g)
    a, b2 = dc.process_ground_turth(gts[0])
    assert torch.equal(a, f2f[0]) and torch.equal(b2, f2g[0])
    dc.check()
    bad = gts.clone()
    bad[0, 1, 0] = float("nan")
    dc({'gts': bad})
    with pytest.raises(ValueError):
        dc.check()
