"""Device-side mirror of `deeplio.common.laserscan.LaserScan` (laserscan.py:12-248): same
constructor, attributes and method names, but the scan lives in HBM and the projection / normal
estimation are gfx950 kernels (csrc/projection.hip).  File reading (`open_scan`'s
`utils.load_velo_scan`) stays with the caller: hand the [N,4] array to `set_scan`."""
import torch

from . import ops


class LaserScan:
    EXTENSIONS_SCAN = ['.bin', '.txt', '.npy']

    def __init__(self, project=False, H=64, W=1024, fov_up=3.0, fov_down=-25.0, min_depth=1, max_depth=80,
                 device="cuda"):
        self.project = project
        self.proj_H, self.proj_W = H, W
        self.proj_fov_up, self.proj_fov_down = fov_up, fov_down
        self.max_depth, self.min_depth = max_depth, min_depth
        self.device = torch.device(device)
        self.reset()

    def reset(self):
        """laserscan.py:27-60 (empty pixels are 0 in every projected attribute)"""
        d, f32, i32 = self.device, torch.float32, torch.int32
        H, W = self.proj_H, self.proj_W
        self.points = torch.zeros(0, 3, dtype=f32, device=d)
        self.remissions = torch.zeros(0, dtype=f32, device=d)
        self.proj_range = torch.zeros(H, W, dtype=f32, device=d)
        self.unproj_range = torch.zeros(0, dtype=f32, device=d)
        self.proj_xyz = torch.zeros(H, W, 3, dtype=f32, device=d)
        self.proj_remission = torch.zeros(H, W, dtype=f32, device=d)
        self.proj_idx = torch.zeros(H, W, dtype=i32, device=d)
        self.proj_x = torch.zeros(0, dtype=i32, device=d)
        self.proj_y = torch.zeros(0, dtype=i32, device=d)
        self.proj_mask = torch.zeros(H, W, dtype=i32, device=d)
        self.proj_normal = None

    def size(self):
        return self.points.shape[0]

    def __len__(self):
        return self.size()

    def set_scan(self, scan):
        """open_scan (laserscan.py:69-95) minus the file read: scan [N,4] = x,y,z,remission;
        points outside [min_depth, max_depth] are dropped."""
        scan = torch.as_tensor(scan, dtype=torch.float32, device=self.device)
        if scan.dim() != 2 or scan.shape[1] < 4:
            raise ValueError("scan must be [N,4]")
        depth = torch.linalg.vector_norm(scan[:, 0:3], 2, dim=1)
        keep = ~((depth > self.max_depth) | (depth < self.min_depth))
        scan = scan[keep]
        self.set_points(scan[:, 0:3].contiguous(), scan[:, 3].contiguous())

    def set_points(self, points, remissions=None):
        """laserscan.py:97-120"""
        self.reset()
        if not torch.is_tensor(points):
            raise TypeError("Scan should be a torch tensor on the HIP device")
        if remissions is not None and not torch.is_tensor(remissions):
            raise TypeError("Remissions should be a torch tensor")
        self.points = points.to(self.device, torch.float32).contiguous()
        if remissions is not None:
            self.remissions = remissions.to(self.device, torch.float32).contiguous()
        else:
            self.remissions = torch.zeros(points.shape[0], dtype=torch.float32, device=self.device)
        if self.project:
            self.do_range_projection()

    def do_range_projection(self):
        """laserscan.py:122-185"""
        out = ops.scan_project(self.points, self.remissions, self.proj_H, self.proj_W, self.proj_fov_up,
                               self.proj_fov_down)
        for k, v in out.items():
            setattr(self, k, v)

    def do_normal_projection(self):
        """laserscan.py:215-248"""
        self.proj_normal = ops.scan_normals(self.proj_xyz, self.proj_range)
        return self.proj_normal

    def velo_image(self, channels=(0, 1, 2, 3, 4, 5, 6, 7), mean=None, crop_top=0, crop_left=0):
        """Kitti.get_velo_image + the normalised branch of transform_images
        (kitti.py:83-97, :345-364) -> [C,H',W'] on the device."""
        if self.proj_normal is None:
            self.do_normal_projection()
        return ops.velo_image(self.proj_xyz, self.proj_remission, self.proj_normal, self.proj_range,
                              self.max_depth, list(channels), mean, crop_top, crop_left)
