"""TEST INFRASTRUCTURE (oracle) — CPU restatements for the point-cloud densification slice.

Checker for `syn3r_amd/pcd.py` / `csrc/knn.hip` (statistical outlier removal) and `csrc/warp.hip` (flow cycle mask).

What is restated, and what pins it:
  * `remove_statistical_outlier` / `uniform_down_sample` — open3d 0.17.0 (env.yml:22), NOT in /root/reference and not
    installed here.  Call site: model/diffusionGS.py:319-322,333.  Published algorithm (Open3D
    `PointCloud::RemoveStatisticalOutliers`): k-d tree query of the `nb_neighbors` nearest points of every point (the point
    itself is returned, at distance 0), mean of their Euclidean distances; cloud mean and standard deviation with n - 1;
    keep 0 < avg < mean + std_ratio * std.  PARITY UNPINNED against open3d itself (the reference holds no fixture of it);
    the k-d tree here is scipy's cKDTree in float64, checked against brute force.
  * flow cycle mask — the forward / backward consistency test behind FSGS' `generate_corresp_mask(..., dist_thresh=3)`
    (model/diffusionGS.py:377).  FSGS and GMFlow are absent: UNPINNED; formula against formula.
Only tests/ import this module.
"""
from __future__ import annotations

import numpy as np


def uniform_down_sample(points: np.ndarray, colors: np.ndarray, every_k_points: int):
    if every_k_points <= 0:
        raise ValueError("illegal sample rate")
    idx = np.arange(0, points.shape[0], every_k_points)
    return points[idx], colors[idx]


def knn_mean_distance(points: np.ndarray, nb_neighbors: int = 20) -> np.ndarray:
    """Mean distance to the nb_neighbors nearest points, self included, nearest first, float64."""
    from scipy.spatial import cKDTree
    pts = np.asarray(points, dtype=np.float64)
    k = min(nb_neighbors, pts.shape[0])
    d, _ = cKDTree(pts).query(pts, k=k)
    d = d.reshape(pts.shape[0], k)
    out = np.zeros(pts.shape[0])
    for t in range(k):                       # sequential accumulation, nearest first (std::accumulate over the sorted result)
        out = out + d[:, t]
    return out / k


def knn_mean_distance_brute(points: np.ndarray, nb_neighbors: int = 20) -> np.ndarray:
    """The same quantity by exhaustive search with the squared distance accumulated as (dx^2 + dy^2) + dz^2."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    k = min(nb_neighbors, n)
    out = np.empty(n)
    for i in range(n):
        dx, dy, dz = pts[:, 0] - pts[i, 0], pts[:, 1] - pts[i, 1], pts[:, 2] - pts[i, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        best = np.sqrt(np.sort(d2)[:k])
        s = 0.0
        for v in best:
            s += v
        out[i] = s / k
    return out


def remove_statistical_outlier(points: np.ndarray, nb_neighbors: int = 20, std_ratio: float = 3.0, avg=None):
    """-> (inlier indices, avg distances, (mean, std, threshold))."""
    avg = knn_mean_distance(points, nb_neighbors) if avg is None else np.asarray(avg, dtype=np.float64)
    valid = avg > 0
    # open3d divides by `valid_distances`: every point whose neighbour query returned anything (all of them: a query returns
    # at least the point itself), while the sums skip the non-positive averages (PointCloud::RemoveStatisticalOutliers)
    nv = int(avg.shape[0])
    mean = float(np.sum(avg[valid]) / nv)
    sq = float(np.sum((avg[valid] - mean) ** 2))
    std = float(np.sqrt(sq / (nv - 1)))
    thr = mean + std_ratio * std
    ind = np.nonzero(valid & (avg < thr))[0]
    return ind, avg, (mean, std, thr)


def filter_dense_cloud(vertices: np.ndarray, colors_rgba: np.ndarray, target: int = 100000):
    """model/diffusionGS.py:314-334."""
    pts = np.asarray(vertices, dtype=np.float64)
    col = np.asarray(colors_rgba, dtype=np.float64)[:, :3] / 255.0
    pts, col = uniform_down_sample(pts, col, pts.shape[0] // target)
    ind, _, _ = remove_statistical_outlier(pts, 20, 3.0)
    return pts[ind], col[ind]


def read_ply(path: str):
    """Minimal reader of the binary PLY layout `write_point_cloud` emits -> (points float64 [n,3], colours uint8 [n,3])."""
    with open(path, "rb") as f:
        header = b""
        while not header.endswith(b"end_header\n"):
            header += f.readline()
        lines = header.decode("ascii").splitlines()
        assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
        n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        props = [l.split()[1:] for l in lines if l.startswith("property")]
        assert props == [["double", "x"], ["double", "y"], ["double", "z"], ["uchar", "red"], ["uchar", "green"], ["uchar", "blue"]]
        rec = np.frombuffer(f.read(), dtype=[("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
    assert rec.shape[0] == n
    return np.stack([rec["x"], rec["y"], rec["z"]], 1), np.stack([rec["r"], rec["g"], rec["b"]], 1)


def flow_cycle_mask(flow_fw: np.ndarray, flow_bw: np.ndarray, thresh: float = 3.0):
    """flows [2,H,W] float32 -> (mask [H,W] float32, cycle error [H,W] float32, +inf outside), fp32 arithmetic in the
    kernel's operation order."""
    f32 = np.float32
    fw, bw = np.asarray(flow_fw, f32), np.asarray(flow_bw, f32)
    _, H, W = fw.shape
    ys, xs = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    tx, ty = xs + fw[0], ys + fw[1]
    inside = (tx >= 0) & (tx <= W - 1) & (ty >= 0) & (ty <= H - 1)
    txc, tyc = np.where(inside, tx, 0).astype(f32), np.where(inside, ty, 0).astype(f32)
    x0 = np.minimum(txc.astype(np.int64), max(W - 2, 0))
    y0 = np.minimum(tyc.astype(np.int64), max(H - 2, 0))
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    ax, ay = (txc - x0.astype(f32)).astype(f32), (tyc - y0.astype(f32)).astype(f32)
    one = f32(1)
    w00, w01, w10, w11 = (one - ax) * (one - ay), ax * (one - ay), (one - ax) * ay, ax * ay

    def sample(g):
        return ((g[y0, x0] * w00 + g[y0, x1] * w01) + g[y1, x0] * w10) + g[y1, x1] * w11

    ex, ey = fw[0] + sample(bw[0]), fw[1] + sample(bw[1])
    d = np.sqrt((ex * ex + ey * ey).astype(f32)).astype(f32)
    d = np.where(inside, d, np.inf).astype(f32)
    return (d < f32(thresh)).astype(f32), d
