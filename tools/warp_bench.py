"""Geometry kernels at the reference's 576x1024 working size (SURVEY.md §8d 'Synthetic warp'): device time per call
and achieved bytes/s against the algorithmic byte counts of §8d (developer tool)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from syn3r_amd import _lib as L
from syn3r_amd.solver_utils.consistency import consistency_check_with_depth
from syn3r_amd.solver_utils.forward_warp import forward_warp, inverse_warp, inverse_warp_batch

dev = torch.device("cuda", 0)
H, W = 576, 1024
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
depth = (2 + 0.5 * np.sin(xs / 97) + 0.3 * np.cos(ys / 53)).astype(np.float32)
K = np.array([[800, 0, W / 2], [0, 800, H / 2], [0, 0, 1]], np.float32)
T1 = np.eye(4, dtype=np.float32)
T2 = np.eye(4, dtype=np.float32); T2[0, 3], T2[2, 3] = 0.05, 0.02
rgb = np.random.default_rng(0).random((3, H, W), dtype=np.float32)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
img, d, k, p1, p2 = t(rgb), t(depth), t(K), t(T1), t(T2)


def dev_ms(fn, n=20):
    fn(); torch.cuda.synchronize()
    with L.kernel_trace() as tr:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return {k: v[1] / n for k, v in tr.result.items()}


px = H * W
r = dev_ms(lambda: inverse_warp(img, d[None], d[None], p1, p2, k, bandwidth=20))
tot = sum(r.values())
print(f"inverse_warp (W2+C1 fused): {tot * 1e3:.1f} us device ({', '.join(f'{a} {b * 1e3:.1f}' for a, b in r.items())}); "
      f"49 B/px algorithmic -> {49 * px / tot / 1e6:.0f} GB/s")
r = dev_ms(lambda: consistency_check_with_depth(d, p2, k, d, p1, k))
tot = sum(r.values())
print(f"consistency_check_with_depth (C1): {tot * 1e3:.1f} us device; 24 B/px -> {24 * px / tot / 1e6:.0f} GB/s")
poses = torch.stack([p2] * 25)
r = dev_ms(lambda: inverse_warp_batch(img, d, torch.stack([d] * 25), p1, poses, k, bandwidth=20), n=5)
tot = sum(r.values())
print(f"inverse_warp_batch x25 (one pose-selection bucket): {tot * 1e3:.1f} us device -> {tot * 1e3 / 25:.1f} us per warp")
frame = (rgb.transpose(1, 2, 0) * 255).astype(np.float64)
args = (frame, None, depth.astype(np.float64), T1.astype(np.float64), T2.astype(np.float64), K.astype(np.float64), None)
forward_warp(*args)
t0 = time.perf_counter()
for _ in range(5):
    forward_warp(*args)
wall = (time.perf_counter() - t0) / 5
with L.kernel_trace() as tr:
    forward_warp(*args); torch.cuda.synchronize()
tot = sum(v[1] for v in tr.result.values())
print(f"forward_warp (W1, fp64 splat): {tot * 1e3:.1f} us device, {wall * 1e3:.1f} ms wall incl. the numpy<->device copies of its "
      f"numpy interface; 144 B/px -> {144 * px / tot / 1e6:.0f} GB/s")
