"""CPU restatement of the reference orchestrator's per-frame post-processing (test oracle, not product code;
imports nothing from syn3r_amd).

Rows O3 / O4 / O5 of SURVEY.md §8a, following /root/reference/model/diffusionGS.py:
  warp_post_frame        :1447-1483   (loop body of warp_images_bw after the inverse warp)
  warp_images_bw         :1367-1510   (whole loop, on oracle/warp_oracle.py's inverse_warp)
  nearby_consistency     :1300-1361   (consistency_check_from_nearby_images_bw)
  fuse_uncertainty       :821-867     (intensity x geometry confidence, cond-image selection)

PARITY of the cv2 steps is UNPINNED against OpenCV itself (cv2 is not installed here and the reference holds no
fixture for them): `cv2.dilate(mask, ones((5,5)))` is restated from OpenCV's documented semantics — the maximum
over the 5x5 window anchored at its centre, border pixels taking the maximum over the in-image part of the window
(BORDER_CONSTANT with morphologyDefaultBorderValue) — with scipy.ndimage as the implementation, i.e. a second,
independent one next to the product's.  `cv2.resize(INTER_LINEAR)` = pixel-centre-aligned bilinear without
antialiasing, `INTER_NEAREST` = floor(dst * scale) source index.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

from . import warp_oracle as WO


def dilate5x5(mask: np.ndarray) -> np.ndarray:
    """cv2.dilate(mask, np.ones((5,5),np.uint8), iterations=1) on an (H,W) or (H,W,C) array of non-negative values
    (diffusionGS.py:1457-1458)."""
    m = np.asarray(mask, dtype=np.float64)
    size = (5, 5) if m.ndim == 2 else (5, 5, 1)
    # values are >= 0, so a constant border of 0 never wins the maximum: same as ignoring out-of-image pixels
    return ndimage.maximum_filter(m, size=size, mode="constant", cval=0.0).astype(mask.dtype)


def block_mean_pool(x: np.ndarray, h: int = 72, w: int = 128) -> np.ndarray:
    """x.reshape(72,8,128,8).transpose(0,2,1,3).reshape(72,128,64).mean(2)  (diffusionGS.py:1481-1483, 853-855),
    written as an explicit block loop (independent of the reshape idiom)."""
    fy, fx = x.shape[0] // h, x.shape[1] // w
    out = np.empty((h, w), dtype=np.result_type(x.dtype, np.float32))
    for i in range(h):
        for j in range(w):
            out[i, j] = np.mean(x[i * fy:(i + 1) * fy, j * fx:(j + 1) * fx].reshape(-1))
    return out


def resize_linear(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(img, (new_w,new_h), interpolation=cv2.INTER_LINEAR): src = (dst + 0.5) * scale - 0.5, clamped taps."""
    H, W = img.shape[:2]
    sy, sx = H / new_h, W / new_w
    ys = (np.arange(new_h) + 0.5) * sy - 0.5
    xs = (np.arange(new_w) + 0.5) * sx - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0), (xs - x0)
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    a = img.astype(np.float64)
    shp = (new_h, 1) + (1,) * (img.ndim - 2)
    shq = (1, new_w) + (1,) * (img.ndim - 2)
    fy, fx = fy.reshape(shp), fx.reshape(shq)
    top = a[y0c][:, x0c] * (1 - fx) + a[y0c][:, x1c] * fx
    bot = a[y1c][:, x0c] * (1 - fx) + a[y1c][:, x1c] * fx
    return (top * (1 - fy) + bot * fy).astype(img.dtype if img.dtype.kind == "f" else np.float64)


def resize_nearest(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src index = floor(dst * scale)."""
    H, W = img.shape[:2]
    yi = np.minimum(np.floor(np.arange(new_h) * (H / new_h)).astype(int), H - 1)
    xi = np.minimum(np.floor(np.arange(new_w) * (W / new_w)).astype(int), W - 1)
    return img[yi][:, xi]


def warp_post_frame(mask_reproj: np.ndarray, warped_chw: np.ndarray, soft_mask_reproj: np.ndarray, h: int, w: int) -> dict:
    """diffusionGS.py:1447-1483 for one frame.  mask_reproj (H,W) bool/0-1, warped_chw (3,H,W) in [0,255],
    soft_mask_reproj (H,W) confidence."""
    mask = 1 - np.asarray(mask_reproj, dtype=np.float64)
    mask[mask < 0.5] = 0
    mask[mask >= 0.5] = 1
    mask = np.repeat(mask[:, :, np.newaxis] * 255.0, repeats=3, axis=2)
    mask_erosion = np.uint8(dilate5x5(mask))                       # PIL.Image.fromarray(np.uint8(..)) round trip
    mask_erosion_ = mask_erosion / 255.0
    mask_erosion_[mask_erosion_ < 0.5] = 0
    mask_erosion_[mask_erosion_ >= 0.5] = 1
    warped = np.asarray(warped_chw).transpose([1, 2, 0])
    cond_ori = warped / 255.0
    cond = np.asarray(np.uint8(warped * (1 - mask_erosion_)), dtype=np.float32) / 255.0
    me = np.mean(mask_erosion_, axis=-1)
    me = block_mean_pool(me, h, w)
    hard = np.where(me < 0.2, 0.0, 1.0)
    soft = 1 - np.asarray(soft_mask_reproj)
    return dict(ero=mask_erosion_[..., 0].astype(np.uint8), cond=cond, cond_ori=cond_ori, masks=hard.astype(np.float32),
                soft=soft, soft_pool=block_mean_pool(soft, h, w))


def warp_images_bw(K, poses, image_l, image_r, depth_l, depth_r, render_depth, h, w):
    """diffusionGS.py:1367-1510 with the CPU inverse warp: per interior pose, warp the nearer end view (first 12 from
    the left, the rest from the right, :1411-1420) with bandwidth 20 and post-process the frame."""
    n = len(poses) - 2
    out = []
    for i in range(n):
        img, dep, pose_s = (image_l, depth_l, poses[0]) if i < 12 else (image_r, depth_r, poses[-1])
        pose_t = poses[i + 1]
        d = WO.inverse_warp(np.asarray(img, dtype=np.float32).transpose(2, 0, 1), np.asarray(dep, dtype=np.float32)[None],
                            np.asarray(render_depth(pose_t), dtype=np.float32)[None], np.asarray(pose_s, dtype=np.float32),
                            np.asarray(pose_t, dtype=np.float32), np.asarray(K, dtype=np.float32), bandwidth=20)
        out.append(warp_post_frame(d["mask_reproj"], d["warped_img"], d["soft_mask_reproj"], h, w))
    return out


def nearby_consistency(K, poses, images, depths, window_radius: int = 1):
    """diffusionGS.py:1300-1361: every frame against its +-1 neighbours, bandwidth 10.
    Returns (geometric uncertainty list, intensity uncertainty list), (H,W) float32 arrays."""
    n = len(poses)
    um, im = [], []
    for cur in range(n):
        masks, warps = [], []
        for ref in range(cur - window_radius, cur + window_radius + 1):
            if ref == cur or ref < 0 or ref >= n:
                continue
            d = WO.inverse_warp(np.asarray(images[ref], dtype=np.float32).transpose(2, 0, 1),
                                np.asarray(depths[ref], dtype=np.float32)[None], np.asarray(depths[cur], dtype=np.float32)[None],
                                np.asarray(poses[ref], dtype=np.float32), np.asarray(poses[cur], dtype=np.float32),
                                np.asarray(K, dtype=np.float32), bandwidth=10)
            masks.append(np.asarray(d["soft_mask_reproj"], dtype=np.float32))
            warps.append(np.asarray(d["warped_img"], dtype=np.float32))
        conf = np.mean(np.stack(masks), axis=0, dtype=np.float32)
        um.append(1 - conf)
        warped = np.mean(np.stack(warps), axis=0, dtype=np.float32)
        diff = warped - np.asarray(images[cur], dtype=np.float32).transpose(2, 0, 1)
        norm = np.sqrt(np.sum(diff.astype(np.float32) ** 2, axis=0, dtype=np.float32))
        im.append(1 - np.exp(-((norm / np.float32(0.1)) ** 3)))
    return um, im


def fuse_uncertainty(cond_images_ori: np.ndarray, gs_images: np.ndarray, soft_masks_reproj_ori: np.ndarray, h: int, w: int):
    """diffusionGS.py:821-867 ('Prob' diffusion types): returns (masks [n,h,w] float32, cond_image list, uncertainty)."""
    unc_mask = 1 - (cond_images_ori.sum(axis=-1, keepdims=True) > 0)
    intensity_conf = np.exp(-((np.linalg.norm(cond_images_ori - gs_images, axis=-1, keepdims=True)) / 0.5) ** 3) * (1 - unc_mask)
    geo_inten_conf = intensity_conf * (1 - soft_masks_reproj_ori[..., None])
    geo_inten_uncertainty = 1 - geo_inten_conf
    buf = []
    for m in geo_inten_uncertainty:
        buf.append(block_mean_pool(np.mean(m, axis=-1), h, w))
    masks = np.stack(buf).astype(np.float32)
    cond_image = np.where(geo_inten_uncertainty > 0.5, gs_images, cond_images_ori)
    cond_image = [np.clip(im, 0, 1) for im in cond_image]
    return masks, cond_image, geo_inten_uncertainty


def warp_images(K, poses, image_l, image_r, depth_l, depth_r, h, w):
    """diffusionGS.py:1512-1606 (forward-warp variant) on oracle/warp_oracle.forward_warp: per interior pose the splat of
    the nearer end view, hole mask = 1 - splat mask, 5x5 dilation, uint8 masking, (h,H/h,w,W/w) pooling, threshold 0.2."""
    n = len(poses) - 2
    out = []
    for i in range(n):
        img, dep, pose_s = (image_l, depth_l, poses[0]) if i < 12 else (image_r, depth_r, poses[-1])
        warped, mask2, _ = WO.forward_warp(np.asarray(img, dtype=np.float64), None, np.asarray(dep, dtype=np.float64),
                                           np.asarray(pose_s, dtype=np.float64), np.asarray(poses[i + 1], dtype=np.float64),
                                           np.asarray(K, dtype=np.float64), None)
        mask = 1 - mask2.astype(np.float64)
        mask[mask < 0.5] = 0
        mask[mask >= 0.5] = 1
        mask = np.repeat(mask[:, :, np.newaxis] * 255.0, repeats=3, axis=2)
        ero = np.uint8(dilate5x5(mask)) / 255.0
        ero[ero < 0.5] = 0
        ero[ero >= 0.5] = 1
        frame = np.uint8(np.uint8(warped) * (1 - ero))
        pooled = block_mean_pool(np.mean(ero, axis=-1), h, w)
        out.append(dict(cond=np.asarray(frame, dtype=np.float32) / 255.0, masks=np.where(pooled < 0.2, 0.0, 1.0), ero=ero[..., 0]))
    return out
