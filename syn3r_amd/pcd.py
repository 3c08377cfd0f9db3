"""Point-cloud filtering of the dense-view cycle on the device.

The reference hands the dust3r cloud to open3d (`model/diffusionGS.py:312-336`):

    pcd.uniform_down_sample(every_k_points=len(points) // 100000)
       .remove_statistical_outlier(nb_neighbors=20, std_ratio=3.0)   ->  select_by_index(ind)  ->  write_point_cloud(.ply)

open3d (0.17.0, env.yml:22) is not part of the reference tree; its published algorithms are restated here on device
tensors: the index stride is a strided copy, the outlier test is `syn3r_pcd_statistical_outlier` (csrc/knn.hip: exact
k = 20 neighbour search in float64).  `PointCloud` carries `.points` / `.colors` the way the trainer's
`reset_gaussians_from_pcd` reads an open3d cloud (diffusionGS.py:1685-1687).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from . import _lib as L


@dataclass
class PointCloud:
    """`.points` [n,3] float64 and `.colors` [n,3] float64 in [0,1] device tensors (open3d's two Vector3dVectors)."""
    points: torch.Tensor
    colors: torch.Tensor

    def __len__(self) -> int:
        return int(self.points.shape[0])

    @classmethod
    def from_arrays(cls, vertices, colors, device) -> "PointCloud":
        """`vertices` [n,3]; `colors` [n,3+] either uint8-valued (0..255, trimesh's RGBA — `colors[:, :3] / 255.0` at
        diffusionGS.py:316 is the caller's job) or already in [0,1]."""
        dev = torch.device(device)
        pts = torch.as_tensor(np.asarray(vertices), dtype=torch.float64).to(dev).contiguous()
        col = torch.as_tensor(np.asarray(colors), dtype=torch.float64).to(dev)[:, :3].contiguous()
        if pts.dim() != 2 or pts.shape[1] != 3 or col.shape != pts.shape:
            raise ValueError(f"PointCloud: vertices {tuple(pts.shape)} / colours {tuple(col.shape)}")
        return cls(pts, col)

    def uniform_down_sample(self, every_k_points: int) -> "PointCloud":
        """open3d `uniform_down_sample`: points 0, k, 2k, ...; k = 0 is an error there too (a cloud of fewer than 100 000
        points makes the reference's `len // 100000` zero and open3d raises)."""
        if every_k_points <= 0:
            raise ValueError("uniform_down_sample: illegal sample rate (every_k_points must be positive; the reference's "
                             "`len(points) // 100000` needs at least 100 000 points)")
        return PointCloud(self.points[::every_k_points].contiguous(), self.colors[::every_k_points].contiguous())

    def remove_statistical_outlier(self, nb_neighbors: int = 20, std_ratio: float = 3.0) -> Tuple["PointCloud", torch.Tensor]:
        """open3d `remove_statistical_outlier` -> (inlier cloud, inlier indices [m] int64), on the device."""
        keep, _, _ = statistical_outlier(self.points, nb_neighbors, std_ratio)
        ind = torch.nonzero(keep, as_tuple=False).reshape(-1)
        return self.select_by_index(ind), ind

    def select_by_index(self, ind: torch.Tensor) -> "PointCloud":
        return PointCloud(self.points[ind].contiguous(), self.colors[ind].contiguous())


def statistical_outlier(points: torch.Tensor, nb_neighbors: int = 20, std_ratio: float = 3.0):
    """-> (keep [n] bool, avg_dist [n] float64, stats [4] float64 = mean, std, threshold, valid count), device tensors."""
    dev = L.require_gpu(points)
    pts = points.to(torch.float64).contiguous()
    n = pts.shape[0]
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError(f"statistical_outlier: points must be [n,3], got {tuple(pts.shape)}")
    lib = L.load()
    avg = torch.empty(n, dtype=torch.float64, device=dev)
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    stats = torch.empty(4, dtype=torch.float64, device=dev)
    ws = L.workspace(dev, lib.syn3r_pcd_outlier_workspace_bytes(n), "pcd")
    rc = lib.syn3r_pcd_statistical_outlier(L.ptr(pts), n, int(nb_neighbors), float(std_ratio), L.ptr(avg), L.ptr(keep),
                                           L.ptr(stats), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, "syn3r_pcd_statistical_outlier")
    return keep.bool(), avg, stats


def filter_dense_cloud(vertices, colors_rgba, device, target: int = 100000, nb_neighbors: int = 20,
                       std_ratio: float = 3.0) -> PointCloud:
    """model/diffusionGS.py:312-334 on the device: colours `[:, :3] / 255`, stride `len // target`, statistical outlier
    removal, inliers selected."""
    col = torch.as_tensor(np.asarray(colors_rgba), dtype=torch.float64)[:, :3] / 255.0
    pcd = PointCloud.from_arrays(vertices, col, device)
    down = pcd.uniform_down_sample(every_k_points=len(pcd) // target)
    inliers, _ = down.remove_statistical_outlier(nb_neighbors=nb_neighbors, std_ratio=std_ratio)
    return inliers


def write_point_cloud(path: str, pcd: PointCloud) -> None:
    """`o3d.io.write_point_cloud(path, pcd)` for a `.ply` (diffusionGS.py:336): binary little-endian, float64 positions,
    uint8 colours (open3d's default PLY layout for a coloured cloud without normals)."""
    pts = pcd.points.detach().to("cpu", torch.float64).numpy()
    col = np.clip(np.rint(pcd.colors.detach().to("cpu", torch.float64).numpy() * 255.0), 0, 255).astype(np.uint8)
    rec = np.empty(pts.shape[0], dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
    rec["red"], rec["green"], rec["blue"] = col[:, 0], col[:, 1], col[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\ncomment Created by syn3r_amd (open3d PLY layout)\n"
              f"element vertex {pts.shape[0]}\nproperty double x\nproperty double y\nproperty double z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(rec.tobytes())


def flow_cycle_mask(flow_fw: torch.Tensor, flow_bw: torch.Tensor, thresh: float = 3.0, want_dist: bool = False):
    """Forward / backward flow cycle-consistency (`syn3r_flow_cycle_mask`): flows [n,2,H,W] -> mask [n,H,W] fp32 in {0,1}
    (and the cycle error [n,H,W] when `want_dist`)."""
    dev = L.require_gpu(flow_fw, flow_bw)
    fw, bw = flow_fw.to(torch.float32).contiguous(), flow_bw.to(torch.float32).contiguous()
    if fw.dim() != 4 or fw.shape[1] != 2 or bw.shape != fw.shape:
        raise ValueError(f"flow_cycle_mask: flows must both be [n,2,H,W], got {tuple(fw.shape)} / {tuple(bw.shape)}")
    n, _, H, W = fw.shape
    mask = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    dist = torch.empty((n, H, W), dtype=torch.float32, device=dev) if want_dist else None
    rc = L.load().syn3r_flow_cycle_mask(L.ptr(fw), L.ptr(bw), n, H, W, float(thresh), L.ptr(mask), L.ptr(dist), L.stream_ptr(dev))
    L.check(rc, "syn3r_flow_cycle_mask")
    return (mask, dist) if want_dist else mask
