"""Seeded synthetic inputs shared by oracle/gen_golden.py (which feeds them to the
reference) and tests/ (which feed them to the oracle and to the HIP path).
Test infrastructure only.  numpy's PCG64 `default_rng(seed)` streams are stable
across numpy versions, so fixtures store outputs only.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32

SCHED_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="v_prediction", interpolation_type="linear", use_karras_sigmas=True, sigma_min=0.002,
    sigma_max=700.0, timestep_spacing="leading", timestep_type="continuous", steps_offset=1,
)


def _rot_y(deg):
    a = np.deg2rad(deg)
    R = np.eye(4, dtype=np.float64)
    R[0, 0], R[0, 2], R[2, 0], R[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    return R


def _rot_x(deg):
    a = np.deg2rad(deg)
    R = np.eye(4, dtype=np.float64)
    R[1, 1], R[1, 2], R[2, 1], R[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return R


def _depth(H, W, sx=97.0, sy=53.0, base=2.0):
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return (base + 0.5 * np.sin(xs / sx) + 0.3 * np.cos(ys / sy)).astype(f32)


WARP_CASES = {
    # name: (H, W, focal, translation, yaw_deg, pitch_deg, bandwidth, stride, seed, holes)
    "small": (64, 96, 80.0, (0.05, 0.0, 0.02), 0.5, 0.0, 20, (1, 1), 0, False),
    "small_bw10": (48, 80, 60.0, (-0.21, 0.07, 0.05), -3.0, 1.5, 10, (1, 1), 1, True),
    "full": (576, 1024, 800.0, (0.05, 0.0, 0.02), 0.5, 0.0, 20, (9, 11), 0, False),
}


def warp_case(name):
    H, W, focal, t, yaw, pitch, bw, stride, seed, holes = WARP_CASES[name]
    rng = np.random.default_rng(seed)
    img = rng.random((3, H, W), dtype=f32)
    sc = W / 1024.0
    depth = _depth(H, W, 97.0 * sc, 53.0 * sc)
    depth_pseudo = _depth(H, W, 91.0 * sc, 57.0 * sc, base=2.02)
    if holes:  # zero-depth (unrendered) regions in both views
        depth[5:12, 10:30] = 0.0
        depth_pseudo[30:36, 50:70] = 0.0
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], dtype=f32)
    pose1 = (_rot_x(0.3) @ np.eye(4)).astype(f32)
    pose1[:3, 3] = np.array([0.01, -0.02, 0.0], dtype=f32)
    T2 = _rot_y(yaw) @ _rot_x(pitch)
    T2[:3, 3] = np.array(t)
    pose2 = (T2 @ pose1.astype(np.float64)).astype(f32)
    return dict(img=img, depth=depth, depth_pseudo=depth_pseudo, pose1=pose1, pose2=pose2, K=K, bandwidth=bw,
                stride=stride, H=H, W=W)


SCHED_CASES = {
    # name: (h, w, dtype, step_i, lambda_kind, mask_kind, stride, seed)
    "tiny_f32": (10, 18, "float32", 10, "binary", "pixel", 1, 11),
    "tiny_f16": (10, 18, "float16", 50, "uniform", "pixel", 1, 12),
    "chan_mask": (12, 16, "float32", 80, "uniform", "channel", 1, 13),
    "tile_f16": (40, 72, "float16", 3, "binary", "pixel", 3, 14),
    "full_f16": (72, 128, "float16", 42, "binary", "pixel", 5, 15),
}


def sched_case(name, F=25, C=4):
    h, w, dt, step_i, lam_kind, mask_kind, stride, seed = SCHED_CASES[name]
    return sched_inputs(h, w, dt, step_i, lam_kind, mask_kind, stride, seed, F, C)


def sched_inputs(h, w, dt, step_i, lam_kind, mask_kind, stride, seed, F=25, C=4):
    rng = np.random.default_rng(seed)
    # sigma at that step sets the sample scale (x = x0 + sigma * eps)
    sig = karras_sigmas(100)[step_i]
    x0 = rng.standard_normal((1, F, C, h, w)).astype(f32) / f32(5.6)
    sample = (x0 + sig * rng.standard_normal((1, F, C, h, w)).astype(f32)).astype(dt)
    model_output = rng.standard_normal((1, F, C, h, w)).astype(f32).astype(dt)
    cond = (x0[0] + 0.05 * rng.standard_normal((F, C, h, w)).astype(f32)).astype(f32)
    temp_cond = np.stack([np.zeros_like(cond), cond], 0)
    if mask_kind == "pixel":
        m = rng.random((1, F - 2, 1, h, w), dtype=f32)
        mask = np.broadcast_to(m, (1, F - 2, C, h, w)).copy()
    else:
        mask = rng.random((1, F - 2, C, h, w), dtype=f32)
    if lam_kind == "binary":
        lam = (rng.random((100, F)) > 0.5).astype(np.float64)
    else:
        lam = rng.random((100, F))
    lam[:, 0] = 1.0
    lam[:, -1] = 1.0
    return dict(model_output=model_output, sample=sample, temp_cond=temp_cond, mask=mask, lambda_ts=lam,
                step_i=step_i, stride=stride, F=F, C=C, h=h, w=w)


def karras_sigmas(n, sigma_min=0.002, sigma_max=700.0, rho=7.0):
    """Karras schedule, float32, with the trailing zero (scheduling_euler_discrete.py:399-423,370)."""
    ramp = np.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (hi + ramp * (lo - hi)) ** rho
    return np.concatenate([s.astype(f32), np.zeros(1, f32)])


def orch_pose_pairs():
    """Two (start, end) w2c pose pairs for pose_interpolation / compute_dists."""
    out = []
    for seed, (yaw, pitch, t) in enumerate([(12.0, -3.0, (0.4, 0.05, -0.1)), (-35.0, 8.0, (-0.7, 0.2, 0.3))]):
        a = (_rot_x(2.0 * seed) @ np.eye(4))
        a[:3, 3] = [0.1 * seed, -0.05, 0.02]
        b = _rot_y(yaw) @ _rot_x(pitch) @ a
        b[:3, 3] = np.array(t)
        out.append((a.astype(np.float32), b.astype(np.float32)))
    return out


def orch_masks():
    """Uncertainty masks [23,72,128] for search_hypers_v2: low, high and ramped uncertainty."""
    rng = np.random.default_rng(21)
    lo = (0.2 * rng.random((23, 72, 128))).astype(f32)
    hi = (0.5 + 0.5 * rng.random((23, 72, 128))).astype(f32)
    ramp = (rng.random((23, 72, 128)) * np.sin(np.linspace(0.1, np.pi - 0.1, 23))[:, None, None]).astype(f32)
    return [lo, hi, ramp]


ORCH_NEARBY_STRIDE = (9, 11)


def orch_nearby_case(H=576, W=1024, n=5):
    """Five frames at the diffusion resolution for consistency_check_from_nearby_images_bw (diffusionGS.py:1300-1361):
    a camera sliding sideways with a small yaw, smooth per-frame depth maps that nearly agree with each other, and smooth
    textured images whose neighbours differ by a few hundredths (so the intensity confidence exp(-(|d|/0.1)^3) is neither
    0 nor 1 everywhere).  Returns K (3,3) f32, poses [n] (4,4) f32 w2c, images [n] (H,W,3) f32, depths [n] (H,W) f32."""
    rng = np.random.default_rng(57)
    sc = W / 1024.0
    K = np.array([[800.0 * sc, 0, W / 2.0], [0, 800.0 * sc, H / 2.0], [0, 0, 1]], dtype=f32)
    base = (_rot_x(0.3) @ np.eye(4))
    base[:3, 3] = [0.01, -0.02, 0.0]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    poses, images, depths = [], [], []
    for i in range(n):
        T = _rot_y(0.2 * i) @ _rot_x(-0.05 * i)
        T[:3, 3] = [0.02 * i, 0.003 * i, 0.004 * i]
        poses.append((T @ base).astype(f32))
        depths.append(_depth(H, W, (97.0 + 3 * i) * sc, (53.0 - 2 * i) * sc, base=2.0 + 0.01 * i))
        im = np.stack([0.5 + 0.3 * np.sin(xs / ((23.0 + 7 * c) * sc) + 0.05 * i) * np.cos(ys / ((31.0 + 5 * c) * sc))
                       for c in range(3)], axis=-1)
        images.append((im + 0.01 * rng.standard_normal(im.shape)).astype(f32))
    depths[2][40:60, 100:180] = 0.0           # an unrendered hole in the middle frame
    return K, poses, images, depths


def orch_fusion_case(n=4, H=576, W=1024):
    """Inputs of the uncertainty fusion / condition-image selection (diffusionGS.py:821-867) at the diffusion resolution: n interior
    frames.  cond_images_ori: smooth warped images in [0,1] with unrendered holes (all-zero pixels); pseudo_images: the n + 2 renders
    (close to the warps, further away in a band so that both sides of the 0.5 threshold occur); soft_masks_reproj_ori in [0,1];
    the two nearby-consistency lists the block's debugging lines read.  Everything float32 (the reference mixes float32 / float64)."""
    rng = np.random.default_rng(91)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    pseudo, cond, soft = [], [], []
    for i in range(n + 2):
        im = np.stack([0.5 + 0.35 * np.sin(xs / (29.0 + 5 * c) + 0.3 * i) * np.cos(ys / (37.0 + 3 * c)) for c in range(3)], axis=-1)
        pseudo.append(im.astype(f32))
    for i in range(n):
        d = 0.02 * rng.standard_normal((H, W, 3)) + 0.6 * np.exp(-((xs - 300.0 - 90 * i) / 60.0) ** 2)[..., None]     # a band that disagrees
        w = np.clip(pseudo[i + 1] + d, 0.0, 1.0).astype(f32)
        w[60 + 40 * i:110 + 40 * i, 500:640] = 0.0                                                                  # holes
        cond.append(w)
        soft.append(np.clip(0.15 + 0.5 * np.sin(xs / 83.0 + i) ** 2 * np.cos(ys / 71.0) ** 2 + 0.05 * rng.random((H, W)), 0, 1).astype(f32))
    soft[n - 1][:] = np.clip(soft[n - 1] + 0.7, 0, 1)          # one frame whose mean uncertainty exceeds 0.6 (the debugging branch)
    near = [rng.random((H, W)).astype(f32) * 0.3 for _ in range(n + 2)]
    near_i = [rng.random((H, W)).astype(f32) * 0.3 for _ in range(n + 2)]
    return dict(pseudo_images=pseudo, cond_images_ori=cond, soft_masks_reproj_ori=soft, nearby=near, nearby_inten=near_i)


def n2_view_poses(V, seed=0):
    """V input-view w2c poses on a gently curving, unevenly spaced path (key-frame selection has something to choose)."""
    rng = np.random.default_rng(100 + seed)
    out = []
    x = 0.0
    for v in range(V):
        x += 0.25 + 0.2 * rng.random()
        p = _rot_y(6.0 * v + 3.0 * rng.standard_normal()) @ _rot_x(2.0 * rng.standard_normal()) @ np.eye(4)
        p[:3, 3] = [x, 0.05 * rng.standard_normal(), 0.1 * np.sin(0.7 * v)]
        out.append(p.astype(np.float32))
    return out


def n2_frame_id(view, k):
    """Marker value of the mock diffused frame k of view pair `view` (frames are [3,4,6] constants)."""
    return (view * 100 + k) / 4096.0


N2_CASES = {
    # name: (V input views, densify_type, fps_keyframe_sampling, num_views_for_pcd_densification)
    "v3_fps4": (3, "interpolate_gs_v2", 1, 4),
    "v3_lin4": (3, "interpolate_gs_v2", 0, 4),
    "v9_fps4": (9, "interpolate_gs_v2", 1, 4),
    "v4_loop0_fps3": (4, "interpolate_loop0_gs", 1, 3),
    "v9_loop0_lin5": (9, "interpolate_loop0_gs", 0, 5),
}


def n2_mask_means(n, seed=0):
    """Mean of the mock correspondence mask per candidate frame (the keep rule thresholds it at 0.3, diffusionGS.py:385)."""
    return np.random.default_rng(200 + seed).random(n).astype(np.float32)


def clip_image(h: int, w: int):
    """Seeded smooth-plus-noise HWC uint8 image, as the orchestrator hands views to the pipeline (oracle/gen_golden.py clip)."""
    import torch
    g = torch.Generator().manual_seed(h)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    img = torch.stack([0.5 + 0.4 * torch.sin(xs / (11 + 3 * c)) * torch.cos(ys / (7 + 2 * c)) for c in range(3)])
    img = (img + 0.1 * torch.rand(3, h, w, generator=g)).clamp(0, 1)
    return (img.permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()


CLIP_CASES = {"a": (576, 1024), "b": (378, 504)}
