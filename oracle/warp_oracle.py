"""numpy restatement of the reference geometry (test oracle, not product code).

Follows
  solver_utils/consistency.py:6-91     get_points_from_depth / transform_points / consistency_check_with_depth
  solver_utils/forward_warp.py:187-279 inverse_warp
  solver_utils/forward_warp.py:7-182   compute_transformed_points / bilinear_splatting / forward_warp
torch's grid_sample (nearest and bilinear, zeros padding, align_corners=False)
is restated with explicit index arithmetic.  float32 for the inverse path,
float64 for the forward splat, as the reference.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _unnormalize(g, size):
    # torch grid_sampler_unnormalize, align_corners=False
    return ((g + f32(1)) * f32(size) - f32(1)) / f32(2)


def _sample_nearest(img, gx, gy):
    """img [C,H,W]; gx, gy [H,W] normalised coords -> [C,H,W] (zeros outside)."""
    C, H, W = img.shape
    sx = np.rint(_unnormalize(gx, W))
    sy = np.rint(_unnormalize(gy, H))
    with np.errstate(invalid="ignore"):
        ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    ix = np.where(ok, sx, 0).astype(np.int64)
    iy = np.where(ok, sy, 0).astype(np.int64)
    out = img[:, iy, ix]
    return np.where(ok[None], out, f32(0)).astype(f32)


def _sample_bilinear(img, gx, gy):
    """img [H,W] -> [H,W]; zeros padding."""
    H, W = img.shape
    sx = _unnormalize(gx, W)
    sy = _unnormalize(gy, H)
    fin = np.isfinite(sx) & np.isfinite(sy)
    sx = np.where(fin, sx, f32(-10)).astype(f32)
    sy = np.where(fin, sy, f32(-10)).astype(f32)
    x0 = np.floor(sx)
    y0 = np.floor(sy)
    wx1 = sx - x0
    wy1 = sy - y0
    wx0 = (x0 + f32(1)) - sx
    wy0 = (y0 + f32(1)) - sy

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xi = np.clip(xx, 0, W - 1).astype(np.int64)
        yi = np.clip(yy, 0, H - 1).astype(np.int64)
        return np.where(ok, img[yi, xi], f32(0)).astype(f32)

    out = tap(y0, x0) * (wx0 * wy0)
    out = out + tap(y0, x0 + 1) * (wx1 * wy0)
    out = out + tap(y0 + 1, x0) * (wx0 * wy1)
    out = out + tap(y0 + 1, x0 + 1) * (wx1 * wy1)
    return np.where(fin, out, f32(0)).astype(f32)


def _apply4(M, x, y, z):
    M = M.astype(f32)
    o = [M[i, 0] * x + M[i, 1] * y + M[i, 2] * z + M[i, 3] for i in range(4)]
    return o


def _apply3(M, x, y, z):
    M = M.astype(f32)
    return [M[i, 0] * x + M[i, 1] * y + M[i, 2] * z for i in range(3)]


def consistency_check_with_depth(depth1, pose1, K1, depth2, pose2, K2):
    """consistency.py:44-91 -> reprojection error (h, w) float32."""
    depth1 = np.asarray(depth1, f32)
    depth2 = np.asarray(depth2, f32)
    pose1 = np.asarray(pose1, f32)
    pose2 = np.asarray(pose2, f32)
    K1 = np.asarray(K1, f32)
    K2 = np.asarray(K2, f32)
    h, w = depth1.shape
    xs, ys = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32), indexing="xy")
    K1inv = np.linalg.inv(K1).astype(f32)
    px, py, pz = _apply3(K1inv, xs, ys, f32(1))                       # :21
    px, py, pz = px * depth1, py * depth1, pz * depth1
    T12 = (pose2 @ np.linalg.inv(pose1)).astype(f32)                  # :37
    T21 = (pose1 @ np.linalg.inv(pose2)).astype(f32)
    with np.errstate(all="ignore"):
        ax, ay, az, aw = _apply4(T12, px, py, pz)
        ax, ay, az = ax / aw, ay / aw, az / aw                        # :38
        ix, iy, iz = _apply3(K2, ax, ay, az)                          # :62
        ix, iy = ix / iz, iy / iz
        gx = ix / (f32(w - 1) / f32(2)) - f32(1)                      # :66-68
        gy = iy / (f32(h - 1) / f32(2)) - f32(1)
        d12 = _sample_bilinear(depth2, gx.astype(f32), gy.astype(f32))  # :71
        bx, by, bz = ax / az * d12, ay / az * d12, az / az * d12      # :74
        cx, cy, cz, cw = _apply4(T21, bx, by, bz)
        cx, cy, cz = cx / cw, cy / cw, cz / cw
        jx, jy, jz = _apply3(K1, cx, cy, cz)                          # :77
        jx, jy = jx / jz, jy / jz
        err = np.sqrt((jx - xs) ** 2 + (jy - ys) ** 2)                # :84
    return err.astype(f32)


def reproj_border_band(depth1, pose1, K1, depth2, pose2, K2, margin=1.0):
    """Pixels of consistency_check_with_depth whose bilinear sample of depth2 (consistency.py:71) has a tap on or
    beyond the zero-padded border (within `margin` px of it): there a 1e-4 px rounding difference in the sample
    position changes the sampled depth by O(depth) and the reprojection error by pixels.  Test helper: the
    comparison of an fp32 implementation with this oracle is bounded hard everywhere else."""
    depth1 = np.asarray(depth1, f32)
    h, w = depth1.shape
    pose1 = np.asarray(pose1, f32)
    pose2 = np.asarray(pose2, f32)
    xs, ys = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32), indexing="xy")
    K1inv = np.linalg.inv(np.asarray(K1, f32)).astype(f32)
    px, py, pz = _apply3(K1inv, xs, ys, f32(1))
    px, py, pz = px * depth1, py * depth1, pz * depth1
    T12 = (pose2 @ np.linalg.inv(pose1)).astype(f32)
    with np.errstate(all="ignore"):
        ax, ay, az, aw = _apply4(T12, px, py, pz)
        ix, iy, iz = _apply3(np.asarray(K2, f32), ax / aw, ay / aw, az / aw)
        gx = (ix / iz) / (f32(w - 1) / f32(2)) - f32(1)
        gy = (iy / iz) / (f32(h - 1) / f32(2)) - f32(1)
        sx, sy = _unnormalize(gx, w), _unnormalize(gy, h)
        m = f32(margin)
        band = (np.abs(sx - 0) <= m) | (np.abs(sx - (w - 1)) <= m) | (np.abs(sy - 0) <= m) | (np.abs(sy - (h - 1)) <= m)
        band |= (np.abs(sx + 1) <= m) | (np.abs(sx - w) <= m) | (np.abs(sy + 1) <= m) | (np.abs(sy - h) <= m)
    return band | ~np.isfinite(sx) | ~np.isfinite(sy)


def inverse_warp(img, depth, depth_pseudo, pose1, pose2, K, bandwidth=20):
    """forward_warp.py:187-279 -> dict of numpy arrays (same keys, bg mask omitted)."""
    img = np.asarray(img, f32)
    depth = np.asarray(depth, f32).reshape(1, *img.shape[1:])
    z = np.asarray(depth_pseudo, f32).reshape(img.shape[1:])
    pose1 = np.asarray(pose1, f32)
    pose2 = np.asarray(pose2, f32)
    K = np.asarray(K, f32)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    _, H, W = img.shape
    ys, xs = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    with np.errstate(all="ignore"):
        x = (xs - cx) / fx
        y = (ys - cy) / fy
        pose = (pose1 @ np.linalg.inv(pose2)).astype(f32)                 # :217
        X, Y, Z, _ = _apply4(pose, x * z, y * z, z)
        x2 = fx * X / Z + cx                                              # :222-223
        y2 = fy * Y / Z + cy
        gx = (f32(2.0) * x2 / f32(W) - f32(1.0)).astype(f32)              # :225
        gy = (f32(2.0) * y2 / f32(H) - f32(1.0)).astype(f32)
        warped_img = _sample_nearest(img, gx, gy)
        warped_depth = _sample_nearest(depth, gx, gy)
        mask_warp = (x2 >= 0) & (x2 < W) & (y2 >= 0) & (y2 < H)          # :233
        dmax = warped_depth.max()                                         # :237
        pos = warped_depth > 0
        dmin = np.where(pos, warped_depth, f32(1e4)).min()                # :239-240
        norm_wd = np.where(pos[0], (warped_depth[0] - dmin) / (dmax - dmin), f32(0)).astype(f32)
        warped_depth = np.where(pos, warped_depth, f32(0)).astype(f32)
        norm_dp = ((z - dmin) / (dmax - dmin)).astype(f32)                # :248
        ad = np.abs(norm_wd - norm_dp)
        mask_depth = ad < f32(0.3)
        mask_depth_strict = ad < f32(0.1)
        mask = mask_warp & mask_depth
        err = consistency_check_with_depth(z, pose2, K, depth[0], pose1, K)   # :259
        mask_reproj = (err < f32(bandwidth)) & mask_warp
        soft = np.exp(-((err / f32(bandwidth)) ** 3)).astype(f32)         # :265
    return {
        "warped_img": warped_img, "warped_depth": warped_depth, "mask_warp": mask_warp, "mask_depth": mask_depth,
        "mask": mask, "warped_masked_img": (warped_img * mask[None]).astype(f32), "mask_inv": ~mask,
        "mask_depth_strict": mask_depth_strict, "mask_reproj": mask_reproj, "soft_mask_reproj": soft,
        "reproj_error": err,
    }


def forward_warp(frame1, mask1, depth1, transformation1, transformation2, intrinsic1, intrinsic2=None):
    """forward_warp.py:141-182 (+ :7-38, :42-127) -> (uint8 (h,w,3), bool (h,w), float64 flow (h,w,2))."""
    frame1 = np.asarray(frame1, np.float64)
    depth1 = np.asarray(depth1, np.float64)
    h, w = depth1.shape
    if mask1 is None:
        mask1 = np.ones((h, w), bool)
    if intrinsic2 is None:
        intrinsic2 = intrinsic1
    T = np.asarray(transformation2, np.float64) @ np.linalg.inv(np.asarray(transformation1, np.float64))
    K1inv = np.linalg.inv(np.asarray(intrinsic1, np.float64))
    K2 = np.asarray(intrinsic2, np.float64)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    ux = K1inv[0, 0] * xs + K1inv[0, 1] * ys + K1inv[0, 2]
    uy = K1inv[1, 0] * xs + K1inv[1, 1] * ys + K1inv[1, 2]
    uz = K1inv[2, 0] * xs + K1inv[2, 1] * ys + K1inv[2, 2]
    wx, wy, wz = depth1 * ux, depth1 * uy, depth1 * uz
    tx = T[0, 0] * wx + T[0, 1] * wy + T[0, 2] * wz + T[0, 3]
    ty = T[1, 0] * wx + T[1, 1] * wy + T[1, 2] * wz + T[1, 3]
    tz = T[2, 0] * wx + T[2, 1] * wy + T[2, 2] * wz + T[2, 3]
    nx = K2[0, 0] * tx + K2[0, 1] * ty + K2[0, 2] * tz
    ny = K2[1, 0] * tx + K2[1, 1] * ty + K2[1, 2] * tz
    nz = K2[2, 0] * tx + K2[2, 1] * ty + K2[2, 2] * tz
    u, v = nx / nz, ny / nz
    flow = np.stack([u - xs, v - ys], axis=2)
    # bilinear_splatting
    ox = (flow[..., 0] + xs) + 1.0
    oy = (flow[..., 1] + ys) + 1.0
    x0 = np.clip(np.floor(ox), 0, w + 1).astype(np.int64)
    y0 = np.clip(np.floor(oy), 0, h + 1).astype(np.int64)
    x1 = np.clip(np.ceil(ox), 0, w + 1).astype(np.int64)
    y1 = np.clip(np.ceil(oy), 0, h + 1).astype(np.int64)
    ox = np.clip(ox, 0, w + 1)
    oy = np.clip(oy, 0, h + 1)
    pnw = (1 - (oy - y0)) * (1 - (ox - x0))
    psw = (1 - (y1 - oy)) * (1 - (ox - x0))
    pne = (1 - (oy - y0)) * (1 - (x1 - ox))
    pse = (1 - (y1 - oy)) * (1 - (x1 - ox))
    logd = np.log(1 + np.clip(nz, 0, 5000))
    dw = np.exp(logd / logd.max() * 50)
    m = mask1.astype(np.float64)
    acc = np.zeros(((h + 2) * (w + 2), 4), np.float64)
    for yy, xx, pw in ((y0, x0, pnw), (y1, x0, psw), (y0, x1, pne), (y1, x1, pse)):
        wgt = pw * m / dw
        lin = (yy * (w + 2) + xx).ravel()
        for k in range(3):
            acc[:, k] += np.bincount(lin, weights=(frame1[..., k] * wgt).ravel(), minlength=acc.shape[0])
        acc[:, 3] += np.bincount(lin, weights=wgt.ravel(), minlength=acc.shape[0])
    acc = acc.reshape(h + 2, w + 2, 4)[1:-1, 1:-1]
    mask2 = acc[..., 3] > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        out = np.where(mask2[..., None], acc[..., :3] / acc[..., 3:4], 0)
    out = np.round(np.clip(out, 0, 255)).astype(np.uint8)
    return out, mask2, flow
