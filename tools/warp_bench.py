"""Geometry kernels at the reference's 576x1024 working size (SURVEY.md §8d 'Synthetic warp'): device time per call
and achieved bytes/s against the algorithmic byte counts of §8d (developer tool)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
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

# orchestrator post-processing: 23 frames at once on the device vs the per-frame host loop
from syn3r_amd import orchestrator as O
poses25 = list(O.pose_interpolation(T1.astype(np.float64), T2.astype(np.float64), num=25))
img_l = (rgb.transpose(1, 2, 0) * 255).astype(np.float32)
depth_dev = d.clone()
run_dev = lambda: O.warp_images_bw_device(K, poses25, img_l, img_l, depth, depth, render_depth=lambda p: depth_dev, h=72, w=128)
r = dev_ms(run_dev, n=3)
print("warp_images_bw_device (23 frames): " + ", ".join(f"{a} {b * 1e3:.0f} us" for a, b in r.items()))
run_dev(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    wd = run_dev()
    m, c, u = O.fuse_uncertainty_device(wd["cond_images_ori"], wd["cond_image"], wd["soft_masks_reproj_ori"])
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
wd1 = inverse_warp(img * 255, d[None], d[None], p1, p2, k, bandwidth=20)
mask2 = wd1["mask_reproj"].cpu().numpy(); warped = wd1["warped_img"].cpu().numpy().transpose([1, 2, 0])
mask = np.repeat((1 - mask2 >= 0.5).astype(np.float64)[:, :, None] * 255.0, 3, axis=2)
ero = (np.uint8(O.dilate5x5(mask)) / 255.0 >= 0.5).astype(np.float64)
cond = np.asarray(np.uint8(warped * (1 - ero)), dtype=np.float32) / 255.0
pooled = O.block_mean_pool(np.mean(ero, axis=-1), 72, 128)
soft = 1 - wd1["soft_mask_reproj"].cpu().numpy(); sp = O.block_mean_pool(soft, 72, 128)
t_host1 = time.perf_counter() - t0
t0 = time.perf_counter()
O.fuse_uncertainty(np.stack([warped / 255.0] * 23), np.stack([cond] * 23), np.stack([soft] * 23))
t_fuse_host = time.perf_counter() - t0
print(f"warps + masks + fusion for one view pair: device path {t_dev * 1e3:.1f} ms wall; per-frame host loop "
      f"{23 * t_host1 * 1e3:.0f} ms + numpy fusion {t_fuse_host * 1e3:.0f} ms")
