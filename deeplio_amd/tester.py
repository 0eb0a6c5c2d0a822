"""Inference / test path: mirror of `deeplio/models/tester.py`.

* `TestStep`     -- the per-batch body of `Tester.test` (tester.py:110-150): eval-mode model on the
                    HIP kernels, `se3_to_SE3` in the tester's quaternion convention, loss; no
                    backward, no optimizer.  Reports the reference's `Inf-Time` (model forward only).
* `local_transform` -- tester.py:188-203: the 4x4 frame-to-frame transform of one sample for
                    `--param xq|x|q|gt`.
* `OdomSeqRes`   -- tester.py:263-326: local -> global integration and the KITTI pose files
                    (`gt_kitti_<date>_<drive>.txt`, `pred_kitti_...txt`, 12 `%.5f` values per row) that
                    `scripts/plot_evo.py` / evo read.  The PNG plot is not reproduced (plotting is out
                    of scope).
File formats and arithmetic (float64 numpy, same operation order) are the reference's, so the
text output is byte-identical for identical local transforms (tests/test_tester.py)."""
import time
import types

import numpy as np
import torch

from . import functional as Fh
from . import losses, misc, nets
from .se3 import se3_to_SE3


def so3_exp(phi):
    """liegroups.torch.SO3.exp(phi).as_matrix() for one rotation vector, float32 like the
    reference's tensors (small-angle branch: first-order, as liegroups does)."""
    phi = np.asarray(phi, np.float32)
    angle = np.float32(np.linalg.norm(phi))
    K = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]], np.float32)
    if np.isclose(angle, 0.):
        return np.eye(3, dtype=np.float32) + K
    axis = phi / angle
    s, c = np.float32(np.sin(angle)), np.float32(np.cos(angle))
    A = K / angle
    return (c * np.eye(3, dtype=np.float32) + (np.float32(1.) - c) * np.outer(axis, axis).astype(np.float32)
            + s * A).astype(np.float32)


def quaternion_to_rotation_matrix(q):
    """deeplio/common/spatial.py quaternion_to_rotation_matrix, q = (w, x, y, z)"""
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def local_transform(pred_t, pred_w, gt_t, gt_w, param="xq"):
    """tester.py:188-203 for one sample (1-D arrays)."""
    T = np.identity(4)
    if param == 'xq':
        T[:3, 3], T[:3, :3] = np.asarray(pred_t), so3_exp(pred_w)
    elif param == 'x':
        T[:3, 3], T[:3, :3] = np.asarray(pred_t), so3_exp(gt_w)
    elif param == 'q':
        T[:3, 3], T[:3, :3] = np.asarray(gt_t), so3_exp(pred_w)
    else:
        T[:3, 3], T[:3, :3] = np.asarray(gt_t), quaternion_to_rotation_matrix(gt_w)
    return T


def global_transform(gt_row):
    """tester.py:171-173 / :181-183: gt row [x(3), R(9), ...] -> 4x4"""
    gt_row = np.asarray(gt_row)
    T = np.identity(4)
    T[:3, 3] = gt_row[0:3]
    T[:3, :3] = gt_row[3:12].reshape(3, 3)
    return T


class OdomSeqRes:
    def __init__(self, date, drive, output_dir="."):
        self.date, self.drive, self.out_dir = date, drive, output_dir
        self.T_local_pred, self.T_global, self.timestamps, self.loss = [], [], [], []

    def add_local_prediction(self, timestamp, loss, T_local, T_gt_global):
        self.timestamps.append(timestamp)
        self.loss.append(loss)
        self.T_local_pred.append(T_local)
        self.T_global.append(T_gt_global)

    def global_predictions(self):
        """tester.py:282-289: T_0i = T_0(i-1) @ T_i"""
        T_0i = self.T_local_pred[0]
        out = [T_0i]
        for T_i in self.T_local_pred[1:]:
            T_0i = np.matmul(T_0i, T_i)
            out.append(T_0i)
        return np.array(out)

    def write_to_file(self):
        T_global = np.array(self.T_global)
        T_pred = self.global_predictions()
        gt_name = "{}/gt_kitti_{}_{}.txt".format(self.out_dir, self.date, self.drive)
        np.savetxt(gt_name, T_global[:, :3, :].reshape(len(T_global), -1), fmt='%.5f', delimiter=' ')
        pred_name = "{}/pred_kitti_{}_{}.txt".format(self.out_dir, self.date, self.drive)
        np.savetxt(pred_name, T_pred[:, :3, :].reshape(len(T_pred), -1), fmt='%.5f', delimiter=' ')
        return gt_name, pred_name


class TestStep:
    """Model + criterion in eval mode; `step()` = tester.py:110-150 for one batch."""
    __test__ = False        # not a pytest class

    def __init__(self, cfg, input_shape, device, batch_size=1):
        self.device = torch.device(device)
        cc = misc.build_config_container(cfg, types.SimpleNamespace(device=str(device), batch_size=batch_size))
        cc = cc if cc is not None else misc.get_config_container()
        if cc.seq_size != 1:
            raise ValueError("Sequence size mus tbe equal 1 in test mode.")      # tester.py:40-42
        Fh.assign_streams(self.device)              # the two encoder streams on hardware queues of their own
        self.model = nets.get_model(input_shape, cfg, self.device)
        self.criterion = losses.get_loss_function(cfg, self.device)
        self.model.eval()
        self.inference_time = 0.0
        self.steps = 0

    # ---- hipGraph replay of the test-mode forward -------------------------------------------
    # At B=1, S=1 the forward is ~150 short launches on three streams: the host needs longer to
    # issue them than the GPU to run them.  Capturing them once (torch.cuda.CUDAGraph = hipGraph on
    # ROCm; the side streams fork/join through events inside the capture) makes a step one launch.
    @torch.no_grad()
    def capture(self, imgs, normals, imus):
        """record the forward for inputs of these shapes; afterwards `forward_graph` replays it"""
        self._static_in = [imgs.clone(), normals.clone(), imus.clone()]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up: workspaces, weight layouts
            for _ in range(3):
                self.model([[self._static_in[0], self._static_in[1]], self._static_in[2]])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_out = self.model([[self._static_in[0], self._static_in[1]], self._static_in[2]])
        return self

    @torch.no_grad()
    def forward_graph(self, imgs, normals, imus):
        for dst, src in zip(self._static_in, (imgs, normals, imus)):
            dst.copy_(src)
        self._graph.replay()
        return self._static_out

    @torch.no_grad()
    def step(self, imgs, normals, imus, gts_f2f, gts_f2g, timed=False):
        if torch.isnan(gts_f2f).any() or torch.isinf(gts_f2f).any():
            raise ValueError("gt-f2f:\n{}".format(gts_f2f))
        if torch.isnan(gts_f2g).any() or torch.isinf(gts_f2g).any():
            raise ValueError("gt-f2g:\n{}".format(gts_f2g))
        if timed:
            torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        pred_t, pred_w = self.model([[imgs, normals], imus])
        if timed:
            torch.cuda.synchronize(self.device)
            self.inference_time += time.perf_counter() - t0
            self.steps += 1
        pred_p, pred_q = se3_to_SE3(pred_t, pred_w, ordering="xyzw")     # tester.py:223-251
        loss = self.criterion(pred_t, pred_w, pred_p, pred_q, gts_f2f[:, :, 0:3], gts_f2f[:, :, 3:],
                              gts_f2g[:, :, 0:3], gts_f2g[:, :, 3:7])
        return pred_t, pred_w, pred_p, pred_q, loss
