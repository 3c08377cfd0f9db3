"""Host-side numerics of the reference orchestrator `model/diffusionGS.py` that sit between the hot
kernels: pose interpolation, warp-mask pooling, uncertainty
fusion and the lambda schedule that feed `svd_render`.  Host code stays Python (north_star); the
per-pixel warps underneath run on the HIP path (`syn3r_amd.solver_utils`).

The reference class itself needs FSGS / cv2 / open3d / trimesh (absent); these functions keep its
method names and argument meaning so that `DiffusionGS` can call them one for one.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
from scipy.interpolate import CubicSpline
from scipy.optimize import minimize_scalar
from scipy.spatial.transform import Rotation as R
from scipy.spatial.transform import Slerp

from . import _lib as L
from .solver_utils.forward_warp import forward_warp, inverse_warp, inverse_warp_batch


def pose_interpolation(pose_start: np.ndarray, pose_end: np.ndarray, num: int = 25) -> np.ndarray:
    """diffusionGS.py:1208-1234 — SLERP rotations + natural cubic spline (2 knots => linear) translations,
    `num` float32 w2c poses."""
    times = np.array([0, num - 1])
    interp_times = np.linspace(0, num - 1, num)
    rotations = R.from_matrix([pose[:3, :3] for pose in [pose_start, pose_end]])
    interpolated_rotations = Slerp(times, rotations)(interp_times)
    translations = np.vstack([pose[:3, 3] for pose in [pose_start, pose_end]])
    interpolated_translations = CubicSpline(times, translations, bc_type="natural")(interp_times)
    return np.array([
        np.vstack([np.hstack([interpolated_rotations[i].as_matrix(), interpolated_translations[i][:, np.newaxis]]),
                   [0, 0, 0, 1]]) for i in range(len(interp_times))]).astype(np.float32)


def compute_dists(interpolated_poses: np.ndarray, type: str = "double_end"):
    """diffusionGS.py:1237-1296 — normalised distance of every pose to its nearer end pose."""
    assert type in ["double_end", "single_end"]
    points = np.array([pp[:3, 3].reshape(1, 3) for pp in interpolated_poses])

    def total_distance(index):
        index = int(index)
        if index <= 0 or index >= len(points) - 1:
            return float("inf")
        left, right = points[:index], points[index:]
        return np.sum(np.linalg.norm(left - left[0], axis=1)) + np.sum(np.linalg.norm(right - right[-1], axis=1))

    result = minimize_scalar(total_distance, bounds=(1, len(points) - 2), method="bounded")
    min_indice = int(result.x)
    if type == "single_end":
        min_indice = len(interpolated_poses) + 10000
    pt1, pt2 = interpolated_poses[0], interpolated_poses[-1]
    diff = np.array([pp[:3, 3] - (pt1 if ii < min_indice else pt2)[:3, 3] for ii, pp in enumerate(interpolated_poses)])
    dists = np.linalg.norm(diff, axis=1)
    return dists / np.max(dists), min_indice


def block_mean_pool(x: np.ndarray, h: int = 72, w: int = 128) -> np.ndarray:
    """The `(72,8,128,8)` reshape/mean idiom (diffusionGS.py:853-855,1481-1483): (h*8, w*8) -> (h, w)."""
    fy, fx = x.shape[0] // h, x.shape[1] // w
    return x.reshape(h, fy, w, fx).transpose(0, 2, 1, 3).reshape(h, w, fy * fx).mean(axis=2)


def dilate5x5(mask: np.ndarray) -> np.ndarray:
    """cv2.dilate(mask, ones((5,5)), iterations=1) (diffusionGS.py:1457-1458): 5x5 max filter, border
    pixels see only the in-image part of the window (OpenCV's default border value for dilation)."""
    t = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32))
    squeeze = t.dim() == 2
    t = t[None, None] if squeeze else t.permute(2, 0, 1)[None]
    out = torch.nn.functional.max_pool2d(torch.nn.functional.pad(t, (2, 2, 2, 2), value=float("-inf")), 5, stride=1)
    return (out[0, 0] if squeeze else out[0].permute(1, 2, 0)).numpy().astype(mask.dtype)


def search_hypers_v2(masks: torch.Tensor, save_path=None, type: str = "double_end", diffusion_steps: int = 100) -> torch.Tensor:
    """diffusionGS.py:1120-1205 — per-frame mean uncertainty -> lambda_ts [steps, F] float64 of {0,1}:
    frame tau follows the warped view while `steps - t > (a u^2 + b u + c) * 100`."""
    assert type in ["double_end", "single_end"]

    def quad_tau_func(u, a=-0.22 / 1.4, b=2.4 * 0.22 / 1.4, c=0.2):
        return (a * u ** 2 + b * u + c) * 100

    m = torch.mean(masks, dim=(-1, -2))
    m = torch.clamp(m / (torch.maximum(m.max(), torch.full_like(m.max(), 0.5))), 0, 1)
    if type == "double_end":
        m = torch.cat([torch.zeros_like(m[0:1]), m, torch.zeros_like(m[:1])], dim=0)
        index_list = list(range(1, m.shape[0] - 1))
    else:
        m = torch.cat([torch.zeros_like(m[0:1]), m], dim=0)
        index_list = list(range(1, m.shape[0]))
    F = m.shape[0]
    lam = np.ones((diffusion_steps, F), dtype=np.float64)
    for t in range(diffusion_steps):
        for tau in index_list:
            lam[t, tau] = 1.0 if diffusion_steps - t > quad_tau_func(m[tau]) else 0.0
    return torch.tensor(lam)


def fuse_uncertainty(cond_images_ori: np.ndarray, gs_images: np.ndarray, soft_masks_reproj_ori: np.ndarray,
                     h: int = 72, w: int = 128):
    """diffusionGS.py:821-862 — intensity confidence exp(-(|warp - gs|_2 / 0.5)^3) * (warp != 0), combined
    with the reprojection confidence; returns (masks [n,h,w] float32 tensor, cond_image [n,H,W,3], uncertainty)."""
    unc_mask = 1 - (cond_images_ori.sum(axis=-1, keepdims=True) > 0)
    intensity_conf = np.exp(-((np.linalg.norm(cond_images_ori - gs_images, axis=-1, keepdims=True)) / 0.5) ** 3) * (1 - unc_mask)
    geo_inten_uncertainty = 1 - intensity_conf * (1 - soft_masks_reproj_ori[..., None])
    masks = np.stack([block_mean_pool(np.mean(u, axis=-1), h, w) for u in geo_inten_uncertainty])
    cond_image = np.where(geo_inten_uncertainty > 0.5, gs_images, cond_images_ori)
    cond_image = [np.clip(im, 0, 1) for im in cond_image]
    return torch.from_numpy(masks).float(), cond_image, geo_inten_uncertainty


def warp_images_bw_device(intrinsics: np.ndarray, interpolated_poses: Sequence[np.ndarray], image_l: np.ndarray,
                          image_r: np.ndarray, depth_l: np.ndarray, depth_r: np.ndarray,
                          render_depth: Callable[[np.ndarray], np.ndarray], device="cuda:0", h: int = 72, w: int = 128):
    """diffusionGS.py:1367-1510 on the device: the interior poses' pseudo-view depths are
    rendered (`render_depth(pose) -> (H,W)` array or tensor = the reference's `render_GS`), the nearer end view is
    inverse-warped into each of them (bandwidth 20) in ONE `syn3r_inverse_warp` call per end view, and
    `syn3r_warp_post` derives, for all frames at once, the hard mask (5x5 dilate, (h,H/h,w,W/w) pooling, threshold
    0.2), the masked uint8-rounded condition images and the soft reprojection uncertainty.  Images are (H,W,3)
    in [0,255], already at the diffusion resolution.  Returns device tensors:
    masks [n,h,w], cond_image [n,H,W,3], masks_ero [n,H,W] u8, soft_masks_reproj [n,h,w],
    soft_masks_reproj_ori [n,H,W], cond_images_ori [n,H,W,3]."""
    dev = torch.device(device)
    n_pose = len(interpolated_poses)
    interp_num = n_pose - 2
    K = torch.tensor(np.asarray(intrinsics), dtype=torch.float32)
    f32 = lambda a: torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a, dtype=torch.float32).to(dev)
    depth_t = torch.stack([f32(render_depth(interpolated_poses[i + 1])) for i in range(interp_num)])
    H, W = depth_t.shape[-2:]
    groups = (("l", image_l, depth_l, interpolated_poses[0], range(0, min(12, interp_num))),       # :1411-1420
              ("r", image_r, depth_r, interpolated_poses[-1], range(min(12, interp_num), interp_num)))
    parts = []
    for _, img, dep, pose_s, idx in groups:
        if len(idx) == 0:
            continue
        out = inverse_warp_batch(f32(img).permute(2, 0, 1).contiguous(), f32(dep).reshape(1, H, W),
                                 depth_t[idx.start:idx.stop], torch.as_tensor(np.asarray(pose_s), dtype=torch.float32),
                                 torch.as_tensor(np.stack([interpolated_poses[i + 1] for i in idx]), dtype=torch.float32), K)
        parts.append(out)
    cat = lambda k: torch.cat([p[k] for p in parts]).contiguous()
    mask_reproj, warped, soft_in = cat("mask_reproj"), cat("warped_img"), cat("soft_mask_reproj")
    n = interp_num
    new = lambda shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
    res = dict(masks=new((n, h, w)), cond_image=new((n, H, W, 3)), masks_ero=new((n, H, W), torch.uint8),
               soft_masks_reproj=new((n, h, w)), soft_masks_reproj_ori=new((n, H, W)), cond_images_ori=new((n, H, W, 3)))
    rc = L.load().syn3r_warp_post(L.ptr(mask_reproj), L.ptr(warped), L.ptr(soft_in), n, H, W, h, w, L.ptr(res["masks_ero"]),
                                  L.ptr(res["cond_image"]), L.ptr(res["cond_images_ori"]), L.ptr(res["soft_masks_reproj_ori"]),
                                  L.ptr(res["masks"]), L.ptr(res["soft_masks_reproj"]), L.stream_ptr(dev))
    L.check(rc, "syn3r_warp_post")
    return res


def warp_images_bw(intrinsics: np.ndarray, interpolated_poses: Sequence[np.ndarray], image_l: np.ndarray,
                   image_r: np.ndarray, depth_l: np.ndarray, depth_r: np.ndarray,
                   render_depth: Callable[[np.ndarray], np.ndarray], device="cuda:0", h: int = 72, w: int = 128):
    """`warp_images_bw_device` with the reference's return shapes (diffusionGS.py:1507-1510): host copies made once,
    after the device work — (image_l/255, image_r/255, masks [n,h,w], cond_image list, aux dict)."""
    d = warp_images_bw_device(intrinsics, interpolated_poses, image_l, image_r, depth_l, depth_r, render_depth,
                              device=device, h=h, w=w)
    ero = d["masks_ero"].cpu().numpy()
    aux = dict(masks_ero=np.repeat(ero[..., None], 3, axis=-1).astype(np.uint8),
               soft_masks_reproj=d["soft_masks_reproj"].cpu(),
               soft_masks_reproj_ori=d["soft_masks_reproj_ori"].cpu().numpy(),
               cond_images_ori=list(d["cond_images_ori"].cpu().numpy()))
    return (image_l / 255.0, image_r / 255.0, d["masks"].cpu().to(torch.float64), list(d["cond_image"].cpu().numpy()), aux)


def warp_images(intrinsics: np.ndarray, interpolated_poses: Sequence[np.ndarray], image_l: np.ndarray,
                image_r: Optional[np.ndarray] = None, depth_l: Optional[np.ndarray] = None,
                depth_r: Optional[np.ndarray] = None, h: int = 72, w: int = 128):
    """diffusionGS.py:1512-1606 — the FORWARD-warp variant (`--interp_type forward_warp`, the constructor default):
    every interior pose receives the nearer end view splatted forward (`syn3r_forward_warp`, fp64, HIP), holes are the
    complement of the splat mask, dilated 5x5, pooled to the latent grid and thresholded at 0.2; the condition image is
    the uint8 warped frame with the dilated holes zeroed.  Images (H,W,3) in [0,255] at the diffusion resolution.
    Returns (image_l/255, image_r/255 or None, masks [n,h,w] float64 tensor, cond_image list) as the reference."""
    n_pose = len(interpolated_poses)
    interp_num = n_pose - 1
    if image_r is not None:
        interp_num -= 1
    K = np.asarray(intrinsics, dtype=np.float64)
    cond_image, masks = [], []
    for i in range(interp_num):
        left = image_r is None or i < 12
        image, depth, pose_s = (image_l, depth_l, interpolated_poses[0]) if left else (image_r, depth_r, interpolated_poses[-1])
        warped, mask2, _ = forward_warp(np.asarray(image, dtype=np.float64), None, np.asarray(depth, dtype=np.float64),
                                        np.asarray(pose_s, dtype=np.float64), np.asarray(interpolated_poses[i + 1], dtype=np.float64),
                                        K, None)
        mask = (1 - mask2.astype(np.float64) >= 0.5).astype(np.float64)
        mask = np.repeat(mask[:, :, None] * 255.0, 3, axis=2)
        ero = np.uint8(dilate5x5(mask)) / 255.0
        ero = (ero >= 0.5).astype(np.float64)
        frame = np.uint8(np.uint8(warped) * (1 - ero))
        cond_image.append(np.asarray(frame, dtype=np.float32) / 255.0)
        pooled = block_mean_pool(np.mean(ero, axis=-1), h, w)
        masks.append(torch.from_numpy((pooled >= 0.2).astype(np.float64)).unsqueeze(0))
    masks = torch.cat(masks)
    return image_l / 255.0, (image_r / 255.0 if image_r is not None else None), masks, cond_image


def fuse_uncertainty_device(cond_images_ori: torch.Tensor, gs_images, soft_masks_reproj_ori: torch.Tensor,
                            h: int = 72, w: int = 128):
    """`fuse_uncertainty` (diffusionGS.py:821-862) in one `syn3r_fuse_uncertainty` call on device tensors
    ([n,H,W,3], [n,H,W,3], [n,H,W]); returns (masks [n,h,w], cond_image [n,H,W,3], uncertainty [n,H,W]), all fp32
    on the device (the numpy restatement above carries the uncertainty in float64: results agree to 1e-6)."""
    dev = L.require_gpu(cond_images_ori, soft_masks_reproj_ori)
    gs = torch.as_tensor(gs_images, dtype=torch.float32).to(dev).contiguous()
    co = cond_images_ori.to(torch.float32).contiguous()
    so = soft_masks_reproj_ori.to(torch.float32).contiguous()
    n, H, W, _ = co.shape
    if gs.shape != co.shape or so.shape != (n, H, W):
        raise ValueError(f"fuse_uncertainty: shapes {tuple(co.shape)} / {tuple(gs.shape)} / {tuple(so.shape)}")
    unc = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    cond = torch.empty_like(co)
    masks = torch.empty((n, h, w), dtype=torch.float32, device=dev)
    rc = L.load().syn3r_fuse_uncertainty(L.ptr(co), L.ptr(gs), L.ptr(so), n, H, W, h, w, L.ptr(unc), L.ptr(cond),
                                         L.ptr(masks), L.stream_ptr(dev))
    L.check(rc, "syn3r_fuse_uncertainty")
    return masks, cond, unc


# ---------------------------------------------------------------------------------------------- O2
def _perturb_interp_pose_candidates(anchor_poses, perturb_num: int = 5, rng=None):
    """diffusionGS.py:653-714 — per anchor: the pose itself + `perturb_num` copies jittered by
    N(0, 0.1 x nearest-neighbour distance) in translation and N(0, 0.1 deg) xyz Euler rotation.
    The reference draws from the unseeded global `np.random`; `rng` (anything with `.normal`) defaults to
    that module so that seeding `np.random` reproduces the reference draw for draw."""
    rng = np.random if rng is None else rng
    translations = np.array([pose[:3, 3] for pose in anchor_poses])
    dists = np.linalg.norm(translations[:, None, :] - translations[None, :, :], axis=-1)
    np.fill_diagonal(dists, np.max(dists))
    nn_dists = np.min(dists, axis=1)
    out = []
    for i, pose in enumerate(anchor_poses):
        group = [anchor_poses[i].astype(np.float32)]
        for _ in range(perturb_num):
            p = pose.copy()
            p[:3, 3] += rng.normal(0, nn_dists[i] * 0.1, size=3)
            rot_noise = R.from_euler("xyz", rng.normal(0, 0.1, size=3) * np.array([1, 1, 1]), degrees=True)
            p[:3, :3] = (rot_noise * R.from_matrix(pose[:3, :3])).as_matrix()
            group.append(p.astype(np.float32))
        out.append(group)
    return out


def _perturb_and_select_interp_poses(anchor_poses, ref_poses, K, render: Callable, perturb_num: int = 5, rng=None,
                                     device="cuda:0"):
    """diffusionGS.py:716-766 — for every anchor pick the candidate whose inverse warp from the nearest
    reference view is MOST uncertain (max mean(1 - soft_mask_reproj), bandwidth 20).
    `render(pose) -> (image (H,W,3) float, depth (H,W) float)` is the reference's `render_GS`.
    All candidates that share a reference view go through ONE batched `syn3r_inverse_warp` call
    (the reference issues 150 separate warps per view pair)."""
    from .solver_utils.forward_warp import inverse_warp_batch
    dev = torch.device(device)
    refs = [render(p) for p in ref_poses]
    ref_t = np.array([p[:3, 3] for p in ref_poses])
    groups = _perturb_interp_pose_candidates(anchor_poses, perturb_num, rng)
    Kt = torch.as_tensor(np.asarray(K), dtype=torch.float32)
    # bucket (group, candidate) by nearest reference view
    buckets: dict = {}
    for gi, group in enumerate(groups):
        for ci, pose in enumerate(group):
            nn = int(np.argmin(np.linalg.norm(ref_t - pose[:3, 3], axis=1)))
            buckets.setdefault(nn, []).append((gi, ci, pose))
    unc = [[0.0] * len(g) for g in groups]
    for nn, items in buckets.items():
        img = torch.as_tensor(refs[nn][0], dtype=torch.float32, device=dev)
        if img.shape[-1] == 3:
            img = img.permute(2, 0, 1).contiguous()
        dep = torch.as_tensor(refs[nn][1], dtype=torch.float32, device=dev)
        dps = torch.stack([torch.as_tensor(render(p)[1], dtype=torch.float32) for _, _, p in items]).to(dev)
        poses = torch.as_tensor(np.stack([p for _, _, p in items]), dtype=torch.float32)
        out = inverse_warp_batch(img, dep, dps, torch.as_tensor(ref_poses[nn], dtype=torch.float32), poses, Kt,
                                 bandwidth=20)
        m = (1 - out["soft_mask_reproj"]).mean(dim=(1, 2)).cpu().numpy()
        for (gi, ci, _), u in zip(items, m):
            unc[gi][ci] = float(u)
    return [group[int(np.argmax(u))] for group, u in zip(groups, unc)]


# ---------------------------------------------------------------------------------------------- O4
def consistency_check_from_nearby_images_bw(intrinsics, interpolated_poses, images, depths, device="cuda:0",
                                            window_radius: int = 1):
    """diffusionGS.py:1300-1361 — every frame is checked against its +-1 neighbours (bandwidth 10):
    geometric uncertainty 1 - mean(soft_mask_reproj), intensity uncertainty
    1 - exp(-(|mean warped - image|_2 / 0.1)^3).  images (H,W,3) float, depths (H,W)."""
    dev = torch.device(device)
    K = torch.as_tensor(np.asarray(intrinsics), dtype=torch.float32)
    imgs = [torch.as_tensor(im, dtype=torch.float32, device=dev).permute(2, 0, 1).contiguous() for im in images]
    deps = [torch.as_tensor(d[None], dtype=torch.float32, device=dev) for d in depths]
    poses = [torch.as_tensor(p, dtype=torch.float32) for p in interpolated_poses]
    uncertainty_masks, intensity_uncertainty_masks = [], []
    n = len(poses)
    for cur in range(n):
        masks, warps = [], []
        for ref in range(cur - window_radius, cur + window_radius + 1):
            if ref == cur or ref < 0 or ref >= n:
                continue
            wd = inverse_warp(imgs[ref], deps[ref], deps[cur], poses[ref], poses[cur], K, bandwidth=10)
            masks.append(wd["soft_mask_reproj"])
            warps.append(wd["warped_img"])
        uncertainty_masks.append(1 - torch.stack(masks).mean(dim=0))
        warped = torch.stack(warps).mean(dim=0)
        intensity_conf = torch.exp(-((torch.norm(warped - imgs[cur], dim=0)) / 0.1) ** 3)
        intensity_uncertainty_masks.append(1 - intensity_conf)
    return uncertainty_masks, intensity_uncertainty_masks


# ---------------------------------------------------------------------------------------------- N2 (key frames, pair graph)
def view_selection_for_pcd_densification(poses: Sequence[np.ndarray], pose_num: int, alpha: float = 1.0, beta: float = 1.0) -> List[int]:
    """diffusionGS.py:185-217 — farthest-point sampling of `pose_num` of the w2c `poses` under the distance
    1 - covisibility,  covisibility(i, j) = exp(-alpha |c_i - c_j|) * exp(-beta angle(z_i, z_j))  (camera centres c and
    viewing directions z of the camera-to-world matrices), seeded with pose 0; each round adds the pose whose nearest
    selected pose is farthest.  Returns the indices in selection order."""
    n = len(poses)
    assert n > pose_num, "The number of poses should be larger than the number of poses to select"
    cam = [np.linalg.inv(p) for p in poses]
    centre = [c[:3, 3] for c in cam]
    axis = [c[:3, 2] for c in cam]
    far = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            gap = np.linalg.norm(centre[i] - centre[j])
            c = np.dot(axis[i], axis[j]) / (np.linalg.norm(axis[i]) * np.linalg.norm(axis[j]))
            turn = np.arccos(np.clip(c, -1.0, 1.0))
            far[i, j] = far[j, i] = 1 - np.exp(-alpha * gap) * np.exp(-beta * turn)
    chosen = [0]
    while len(chosen) < pose_num:
        nearest = far[chosen].min(axis=0)
        nearest[chosen] = -np.inf
        chosen.append(int(np.argmax(nearest)))
    return chosen


def key_frame_template(interpolated_poses: Sequence[np.ndarray], n_frames: int, num_key: int, fps: bool) -> np.ndarray:
    """diffusionGS.py:274-284 — which of a view pair's first n_frames - 1 frames feed the point-cloud densification:
    `num_key` indices by farthest-pose sampling (sorted) or evenly spaced, the LAST of them dropped (the pair's end
    frame is the next pair's start frame), as a boolean template of length n_frames - 1."""
    if fps:
        key = sorted(view_selection_for_pcd_densification(interpolated_poses, num_key, alpha=1.0, beta=1.0))
    else:
        key = list(np.linspace(0, n_frames - 1, num_key, dtype=int))
    template = np.zeros(n_frames - 1, dtype=bool)
    template[np.asarray(key[:-1], dtype=int)] = True
    return template


def complete_pair_graph(global_image_inds: Sequence[int]) -> List[tuple]:
    """dust3r `make_pairs(scene_graph='complete')` over the kept key frames (diffusionGS.py:401): every unordered pair
    once, as global frame indices.  Used when the injected dust3r object has no `make_pairs` of its own."""
    g = list(global_image_inds)
    return [(g[a], g[b]) for a in range(len(g)) for b in range(a + 1, len(g))]
