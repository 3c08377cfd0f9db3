"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by syn3r_amd/): CPU restatement of `distCUDA2`, the mean squared
distance of every point to its 3 nearest neighbours that FSGS' `GaussianModel.create_from_pcd` uses to initialise the
Gaussian scales (call site model/diffusionGS.py:1685-1687 -> reset_gaussians_from_pcd; the `simple-knn` CUDA
extension is an un-vendored submodule).

PARITY UNPINNED against the CUDA extension (absent).  The quantity itself is unambiguous: any exact 3-nearest-
neighbour search yields the same three distances, so the oracle is (a) a brute-force search in the HIP kernel's own
fp32 operation order - d2 = (dx*dx + dy*dy) + dz*dz, mean = ((b0 + b1) + b2) / 3 with b0 <= b1 <= b2 - which the
kernel must match BIT FOR BIT, and (b) scipy's k-d tree in float64 as an independent algorithm."""
import numpy as np


def mean_dist2_bruteforce(points: np.ndarray, chunk: int = 2048) -> np.ndarray:
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, chunk):
        q = p[s:s + chunk]
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz                       # fp32, the kernel's order, no fused multiply-add
        d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf     # a point is not its own neighbour
        b = np.sort(np.partition(d2, 2, axis=1)[:, :3], axis=1)
        out[s:s + chunk] = ((b[:, 0] + b[:, 1]) + b[:, 2]) / np.float32(3.0)
    return out


def mean_dist2_kdtree(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import cKDTree
    p = np.asarray(points, dtype=np.float64)
    d, _ = cKDTree(p).query(p, k=4)
    return (d[:, 1:] ** 2).mean(1)
