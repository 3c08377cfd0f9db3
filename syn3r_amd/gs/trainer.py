"""HOT LOOP A host side: the trainer surface `DiffusionGS` consumes
(`model/diffusionGS.py:51,132-139,154,166,1612-1641,1685-1697`), implemented on the HIP rasteriser.

The reference delegates all of this to `FSGS.utils.trainer_v4.GSTrainer` (un-vendored submodule): only the
call surface is visible.  This module provides that surface with the published 3DGS training step
(render -> (1-lambda) L1 + lambda (1-SSIM) photometric loss weighted by the camera confidence -> backward ->
Adam; the loss and the Adam update are the fused HIP operators of `train_ops.py`), no densification
heuristics (out of scope).  Camera conventions follow the published 3DGS/FSGS code:
`world_view_transform` and `full_proj_transform` are the TRANSPOSED matrices.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib as L
from ..raster import GaussianRasterizationSettings, GaussianRasterizer
from .train_ops import FusedAdam, image_metrics, knn3_mean_dist2, l1_loss, photometric_loss

SH_C0 = 0.28209479177387814


def _world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    Rt = np.zeros((4, 4), dtype=np.float32)
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt


def _projection(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = znear / right
    P[1, 1] = znear / top
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class Camera:
    """`FSGS.scene.cameras.Camera(colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask, image_name, uid, trans, scale,
    data_device, cam_confidence)` as constructed at diffusionGS.py:161-163.  R is camera-to-world rotation and T the
    world-to-camera translation (COLMAP/3DGS convention)."""

    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask=None, image_name="", uid=0,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, data_device="cuda", cam_confidence: float = 1.0):
        self.uid, self.colmap_id, self.image_name = uid, colmap_id, image_name
        self.R, self.T, self.FoVx, self.FoVy = np.asarray(R, np.float32), np.asarray(T, np.float32), FoVx, FoVy
        self.cam_confidence = float(cam_confidence)
        self.is_pseudo = False           # set by GSTrainer.update_cameras for SVD pseudo-views
        self.data_device = torch.device(data_device)
        self.original_image = None
        if image is not None:
            self.original_image = torch.as_tensor(image, dtype=torch.float32).clamp(0.0, 1.0).to(self.data_device)
            self.image_height, self.image_width = self.original_image.shape[1:]
        self.zfar, self.znear = 100.0, 0.01
        w2c = torch.tensor(_world2view(self.R, self.T))
        self.world_view_transform = w2c.transpose(0, 1).to(self.data_device)
        self.projection_matrix = _projection(self.znear, self.zfar, FoVx, FoVy).transpose(0, 1).to(self.data_device)
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = self.world_view_transform.inverse()[3, :3]

    @classmethod
    def from_w2c(cls, w2c: np.ndarray, K: np.ndarray, H: int, W: int, image=None, **kw):
        """Build from a 4x4 world-to-camera matrix and intrinsics (the orchestrator's pose format)."""
        fovx = 2 * math.atan(W / (2 * K[0, 0]))
        fovy = 2 * math.atan(H / (2 * K[1, 1]))
        cam = cls(0, np.asarray(w2c[:3, :3]).T, np.asarray(w2c[:3, 3]), fovx, fovy, image, **kw)
        if image is None:
            cam.image_height, cam.image_width = H, W
        return cam

    def get_image(self):
        return self.original_image

    def get_calib_matrix_nerf(self):
        """(K, w2c) as diffusionGS.py:67-70 reads them."""
        fx = self.image_width / (2 * math.tan(self.FoVx / 2))
        fy = self.image_height / (2 * math.tan(self.FoVy / 2))
        K = torch.tensor([[fx, 0, self.image_width / 2], [0, fy, self.image_height / 2], [0, 0, 1]], dtype=torch.float32)
        return K, self.world_view_transform.transpose(0, 1).cpu()


def build_rotation(q: torch.Tensor) -> torch.Tensor:
    """Rotation matrices [n,3,3] of (w, x, y, z) quaternions, normalised first (published 3DGS general_utils)."""
    q = q / torch.norm(q, dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.empty((q.shape[0], 3, 3), dtype=q.dtype, device=q.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class GaussianModel:
    """Trainable Gaussian parameters with the published activations (exp scale, sigmoid opacity, unit quaternion)."""

    def __init__(self, xyz, log_scales, rotations, opacity_logits, shs, sh_degree: int = 3, device="cuda"):
        dev = torch.device(device)
        p = lambda t: torch.nn.Parameter(torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous())
        self._xyz, self._scaling, self._rotation = p(xyz), p(log_scales), p(rotations)
        self._opacity, self._features = p(opacity_logits), p(shs)
        self.confidence = torch.ones(self._xyz.shape[0], device=dev)
        self.max_sh_degree = self.active_sh_degree = sh_degree

    def parameters(self):
        return [self._xyz, self._features, self._opacity, self._scaling, self._rotation]

    @property
    def get_xyz(self): return self._xyz
    @property
    def get_scaling(self): return torch.exp(self._scaling)
    @property
    def get_rotation(self): return torch.nn.functional.normalize(self._rotation)
    @property
    def get_opacity(self): return torch.sigmoid(self._opacity)
    @property
    def get_features(self): return self._features

    # ---- densification statistics (published 3DGS GaussianModel: xyz_gradient_accum / denom / max_radii2D)
    def ensure_stats(self):
        n, dev = self._xyz.shape[0], self._xyz.device
        if getattr(self, "xyz_gradient_accum", None) is None or self.xyz_gradient_accum.shape[0] != n:
            self.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
            self.denom = torch.zeros(n, 1, device=dev)
            self.max_radii2D = torch.zeros(n, device=dev)

    @torch.no_grad()
    def add_densification_stats(self, viewspace_grad: torch.Tensor, update_filter: torch.Tensor, radii: Optional[torch.Tensor] = None):
        """Accumulate the norm of the screen-space positional gradient of the visible Gaussians (3DGS section 5.2: the
        quantity the clone / split decision thresholds) and their largest screen radius."""
        self.ensure_stats()
        if (radii is not None and update_filter is None and viewspace_grad.is_cuda and viewspace_grad.dtype == torch.float32
                and viewspace_grad.is_contiguous() and radii.dtype == torch.int32):
            # the training loop's case (visible = radii > 0): one HIP launch, no boolean-mask indexing (each of which is a
            # `nonzero` with a device -> host synchronisation)
            L.check(L.load().syn3r_densification_stats(int(radii.shape[0]), L.ptr(radii), L.ptr(viewspace_grad),
                                                       L.ptr(self.xyz_gradient_accum), L.ptr(self.denom), L.ptr(self.max_radii2D),
                                                       L.stream_ptr(viewspace_grad.device)), "densification_stats")
            return
        if update_filter is None:
            update_filter = radii > 0
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1
        if radii is not None:
            self.max_radii2D[update_filter] = torch.max(self.max_radii2D[update_filter], radii[update_filter].to(self.max_radii2D.dtype))

    def capture(self) -> dict:
        """Parameter tensors of a checkpoint (published 3DGS `GaussianModel.capture`, reduced to what this model holds)."""
        return dict(active_sh_degree=self.active_sh_degree, xyz=self._xyz.detach().clone(),
                    features=self._features.detach().clone(), scaling=self._scaling.detach().clone(),
                    rotation=self._rotation.detach().clone(), opacity=self._opacity.detach().clone(),
                    confidence=self.confidence.clone())

    def restore(self, state: dict):
        dev = self._xyz.device
        p = lambda t: torch.nn.Parameter(t.to(dev, torch.float32).contiguous())
        self._xyz, self._features, self._scaling = p(state["xyz"]), p(state["features"]), p(state["scaling"])
        self._rotation, self._opacity = p(state["rotation"]), p(state["opacity"])
        self.confidence = state["confidence"].to(dev)
        self.active_sh_degree = int(state["active_sh_degree"])

    def set_from_pcd(self, points: np.ndarray, colors: np.ndarray, append: bool):
        """Published 3DGS `create_from_pcd`: DC colour = (rgb - 0.5) / C0, higher SH zero, isotropic scale =
        sqrt(mean squared distance to the 3 nearest neighbours), identity rotation, opacity 0.1.  The reference does this
        inside FSGS (`simple-knn` CUDA extension, absent); the exact 3-NN search is `csrc/knn.hip` (`train_ops.knn3_mean_dist2`)."""
        dev = self._xyz.device
        to_dev = lambda a: (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(dev, torch.float32)
        pts, rgb = to_dev(points), to_dev(colors)
        if pts.dim() != 2 or pts.shape[1] != 3 or rgb.shape != pts.shape:
            raise ValueError("reset_gaussians_from_pcd: points and colours must both be [n,3]")
        n = pts.shape[0]
        if n >= 4:
            d2 = knn3_mean_dist2(pts)                 # HIP: exact 3-NN (csrc/knn.hip), the simple-knn quantity
        else:                                         # degenerate clouds (fewer than 3 neighbours): mean over what there is
            d = torch.cdist(pts, pts) ** 2
            d2 = d.topk(n, dim=1, largest=False).values[:, 1:].mean(1) if n > 1 else torch.full((n,), 1e-4, device=dev)
        scales = torch.log(torch.sqrt(d2.clamp_min(1e-7)))[:, None].repeat(1, 3)
        M = self._features.shape[1]
        feats = torch.zeros(n, M, 3, device=dev)
        feats[:, 0] = (rgb - 0.5) / SH_C0
        rots = torch.zeros(n, 4, device=dev)
        rots[:, 0] = 1.0
        opac = torch.full((n,), math.log(0.1 / 0.9), device=dev)
        conf = torch.ones(n, device=dev)
        cat = (lambda old, new: torch.cat([old.detach(), new])) if append else (lambda old, new: new)
        p = lambda t: torch.nn.Parameter(t.contiguous())
        self._xyz, self._features = p(cat(self._xyz, pts)), p(cat(self._features, feats))
        self._scaling, self._rotation = p(cat(self._scaling, scales)), p(cat(self._rotation, rots))
        self._opacity = p(cat(self._opacity, opac))
        self.confidence = cat(self.confidence, conf)

    def to(self, device):
        """diffusionGS.py:901-907 moves the Gaussians off and back on the GPU around svd_render; with 288 GB of HBM
        nothing has to move — kept as a no-op-compatible method."""
        dev = torch.device(device)
        for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"):
            t = getattr(self, name)
            t.data = t.data.to(dev)
        self.confidence = self.confidence.to(dev)
        return self


@dataclass
class OptimizationParams:
    iterations: int = 10_000
    position_lr: float = 1.6e-4
    feature_lr: float = 2.5e-3
    opacity_lr: float = 5e-2
    scaling_lr: float = 5e-3
    rotation_lr: float = 1e-3
    lambda_dssim: float = 0.2          # published 3DGS: L = (1 - lambda) L1 + lambda (1 - SSIM)
    pseudo_cam_sampling_rate: float = 0.02
    seed: int = 0
    use_lpips_loss: bool = False       # toggled by DiffusionGS.run (diffusionGS.py:1690,1697); see GSTrainer.train_step
    lpips_weight: float = 0.0
    # adaptive density control, published 3DGS defaults (Kerbl et al. 2023 section 5.2 and its released arguments); used when
    # training()/finetune() run with disable_densification=False
    percent_dense: float = 0.01
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_grad_threshold: float = 0.0002
    prune_min_opacity: float = 0.005
    prune_screen_size: float = 20.0


class _Scene:
    """The part of FSGS' `Scene` the orchestrator touches: `model_path` (checkpoint directory, diffusionGS.py:1611),
    `train_cameras` ({resolution scale: [Camera]}; backed up and restored around a finetune, :1627,1641) and
    `getTrainCameras()`.  Pseudo-views appended by `update_cameras` carry `is_pseudo`; `getTrainCameras` returns the
    real input views, `getPseudoCameras` the appended ones."""

    def __init__(self, cams, model_path: Optional[str] = None):
        self.train_cameras: Dict[float, List[Camera]] = {1.0: list(cams)}
        self.model_path = model_path

    def getTrainCameras(self, scale: float = 1.0, ordered: bool = False):
        return [c for c in self.train_cameras[scale] if not getattr(c, "is_pseudo", False)]

    def getPseudoCameras(self, scale: float = 1.0):
        return [c for c in self.train_cameras[scale] if getattr(c, "is_pseudo", False)]


class GSTrainer:
    def __init__(self, gaussians: GaussianModel, train_cameras: Sequence[Camera], opt: Optional[OptimizationParams] = None,
                 background=(0.0, 0.0, 0.0), model_path: Optional[str] = None,
                 checkpoint_iterations: Optional[Sequence[int]] = None):
        self.gaussians, self.opt = gaussians, opt or OptimizationParams()
        self.scene = _Scene(train_cameras, model_path)
        self.dust3r = None                # injected: to(device) / [make_pairs] / run(frames, c2w_poses=, intrinsics=, preset_pairs=)
        self.flow_net = None              # injected: flow_net(image_a [3,H,W], image_b [3,H,W]) -> flow a->b [2,H,W] (GMFlow's role)
        self.lpips = None                 # injected: syn3r_amd.gs.lpips.LPIPS with loaded weights (the `lpips` package's role)
        self.checkpoint_iterations: List[int] = list(checkpoint_iterations or [])
        self.iteration = 0
        self.densify = False              # adaptive density control inside train_step (training / finetune set it)
        self.truncated_renders = 0        # renders whose (Gaussian, tile) pair list outgrew the async capacity (see _loop)
        self.background = torch.tensor(background, dtype=torch.float32, device=gaussians._xyz.device)
        self._rng = np.random.default_rng(self.opt.seed)
        self.reset_optimizers()

    # ------------------------------------------------------------------ surface used by DiffusionGS
    def reset_optimizers(self):
        g, o = self.gaussians, self.opt
        self.optimizer = FusedAdam([
            {"params": [g._xyz], "lr": o.position_lr}, {"params": [g._features], "lr": o.feature_lr},
            {"params": [g._opacity], "lr": o.opacity_lr}, {"params": [g._scaling], "lr": o.scaling_lr},
            {"params": [g._rotation], "lr": o.rotation_lr}], eps=1e-15)

    def reset_gs(self):
        return None

    @property
    def pseudo_cameras(self) -> List[Camera]:
        return self.scene.getPseudoCameras()

    def update_cameras(self, views, poses, K, cam_confidences, append: bool = True, load_iteration=None):
        """diffusionGS.py:1631 — register SVD pseudo-views ([3,H,W] tensors + w2c poses) with their confidence in
        `scene.train_cameras` (appended, as the reference; the orchestrator restores the list after the finetune)."""
        if np.isscalar(cam_confidences):
            cam_confidences = [float(cam_confidences)] * len(views)
        cams = [Camera.from_w2c(np.asarray(p), np.asarray(K), v.shape[1], v.shape[2], image=v, cam_confidence=c,
                                data_device=self.gaussians._xyz.device) for v, p, c in zip(views, poses, cam_confidences)]
        for c in cams:
            c.is_pseudo = True
        for scale, lst in self.scene.train_cameras.items():
            keep = list(lst) if append else [c for c in lst if not getattr(c, "is_pseudo", False)]
            self.scene.train_cameras[scale] = keep + cams

    # ---- checkpoints with the reference's file names (diffusionGS.py:1611-1625)
    def _ckpt_dir(self) -> str:
        if not self.scene.model_path:
            raise RuntimeError("GSTrainer: scene.model_path is not set (checkpoints need a directory)")
        os.makedirs(self.scene.model_path, exist_ok=True)
        return self.scene.model_path

    def save_checkpoint(self, iteration: int, refine_epoch: Optional[int] = None, latest: bool = False) -> str:
        """`chkpnt{iteration}.pth` after the initial training, `refine_{epoch}_chkpnt{iteration}.pth` after a finetune,
        `chkpnt_latest.pth` as the fallback the reference looks for: a `(capture, iteration)` tuple as published 3DGS."""
        name = "chkpnt_latest.pth" if latest else (f"chkpnt{iteration}.pth" if refine_epoch is None
                                                   else f"refine_{refine_epoch}_chkpnt{iteration}.pth")
        path = os.path.join(self._ckpt_dir(), name)
        torch.save((self.gaussians.capture(), int(iteration)), path)
        return path

    def load_checkpoint(self, checkpoint: str):
        """diffusionGS.py:1618,1624 — restore the Gaussians from a checkpoint file; optimiser state starts afresh
        (the orchestrator calls reset_optimizers right after, :1634).  The file is THIS repository's format — a
        `(dict of tensors / ints, iteration)` pair written by `save_checkpoint` under the reference's file names — and
        is read with the tensors-only loader (`weights_only=True`: no pickled code is executed).  FSGS' own
        `chkpnt*.pth` (a tuple of optimiser and model objects) is a different format and is rejected by that loader."""
        state, it = torch.load(checkpoint, map_location=self.gaussians._xyz.device, weights_only=True)
        self.gaussians.restore(state)
        self.iteration = int(it)
        self.reset_optimizers()

    def reset_gaussians_from_pcd(self, pcd, append_to_old_gaussians: bool = False):
        """diffusionGS.py:1685-1687 — re-initialise (or extend) the Gaussians from a dense point cloud: an object with
        `.points` / `.colors` (open3d) or a (points [n,3], colours [n,3] in [0,1]) pair."""
        if hasattr(pcd, "points"):
            pts, col = pcd.points, pcd.colors            # open3d-like cloud or syn3r_amd.pcd.PointCloud (device tensors)
        else:
            pts, col = pcd
        self.gaussians.set_from_pcd(pts, col, append=append_to_old_gaussians)
        self.reset_optimizers()

    def generate_corresp_mask(self, gs_renderings, svd_outputs, dist_thresh: float = 3, desc_only: bool = False):
        """diffusionGS.py:377 — per (Gaussian render, diffused frame) pair the mask of pixels with a consistent dense
        correspondence, and the flows.  FSGS' wrapper and its GMFlow network are absent; here the flow network is the
        injected attribute `flow_net` (called once per direction) and the test is the forward / backward cycle error below
        `dist_thresh` pixels (`syn3r_flow_cycle_mask`, csrc/warp.hip).  Images go to `flow_net` exactly as the orchestrator
        passes them ([3,H,W] tensors; the reference hands over the render in [0,1] and the diffused frame in [0,255]).
        Returns (masks, flow_bwfw): masks[i] is [1,H,W] fp32 in {0,1} (the caller reads `masks[0][0]`), flow_bwfw[i] the
        (forward, backward) flow pair."""
        if desc_only:
            raise NotImplementedError("desc_only=True (descriptor matching) is internal to FSGS; the orchestrator passes False")
        if self.flow_net is None:
            raise RuntimeError("generate_corresp_mask needs GSTrainer.flow_net (the optical-flow network, GMFlow in the reference)")
        assert len(gs_renderings) == len(svd_outputs)
        from ..pcd import flow_cycle_mask
        dev = self.gaussians._xyz.device
        masks, flows = [], []
        for a, b in zip(gs_renderings, svd_outputs):
            a, b = a.to(dev, torch.float32), b.to(dev, torch.float32)
            fw = torch.as_tensor(self.flow_net(a, b), dtype=torch.float32).to(dev)
            bw = torch.as_tensor(self.flow_net(b, a), dtype=torch.float32).to(dev)
            masks.append(flow_cycle_mask(fw[None], bw[None], float(dist_thresh)))
            flows.append((fw, bw))
        return masks, flows

    def find_nearest_cam(self, cams: Sequence[Camera], candidates: Sequence[Camera], multi_view_max_angle: float = 30.0,
                         multi_view_min_dis: float = 0.01, multi_view_max_dis: float = 1.5, multi_view_num: int = 8):
        """diffusionGS.py:475-477 (only reached from the dead `_extrapolate_from_gs`; FSGS source absent, so this follows
        the public multi-view neighbour selection it is named after — UNPINNED): for every camera the candidates
        whose viewing direction is within `multi_view_max_angle` degrees and whose centre is within
        [min_dis, max_dis], nearest first; stored as `cam.nearest_id` and returned."""
        out = []
        c_pos = np.stack([c.camera_center.detach().cpu().numpy() for c in candidates]) if len(candidates) else np.zeros((0, 3))
        c_dir = np.stack([np.asarray(c.R)[:, 2] for c in candidates]) if len(candidates) else np.zeros((0, 3))
        for cam in cams:
            pos, d = cam.camera_center.detach().cpu().numpy(), np.asarray(cam.R)[:, 2]
            dis = np.linalg.norm(c_pos - pos[None], axis=1)
            cosang = np.clip(c_dir @ d / (np.linalg.norm(c_dir, axis=1) * np.linalg.norm(d) + 1e-12), -1, 1)
            ang = np.degrees(np.arccos(cosang))
            ok = (ang < multi_view_max_angle) & (dis > multi_view_min_dis) & (dis < multi_view_max_dis)
            order = [int(i) for i in np.argsort(dis) if ok[i]][:multi_view_num]
            cam.nearest_id = order
            out.append(order)
        return out

    def evaluate(self, cams: Optional[Sequence[Camera]] = None) -> dict:
        """PSNR / SSIM of the current Gaussians over held-out (or the training) cameras, computed on the device
        (`train_ops.image_metrics`): the psnr / ssim columns of the per-scene record (SURVEY.md §8e;
        scripts/summarize_dl3dv.py:11-80 tabulates them).  LPIPS is reported when the trainer holds an `lpips` model
        (weights supplied by the caller), NaN otherwise."""
        cams = list(cams) if cams is not None else self.scene.getTrainCameras()
        ps, ss, ls = [], [], []
        with torch.no_grad():
            for cam in cams:
                img = self.render_view(cam)["render"].clamp(0, 1)
                m = image_metrics(img, cam.original_image)
                ps.append(m[0])
                ss.append(m[1])
                if self.lpips is not None:
                    ls.append(self.lpips(img, cam.original_image))
        p, s_ = torch.stack(ps).mean(), torch.stack(ss).mean()
        lp = float(torch.stack(ls).mean()) if ls else float("nan")
        return {"psnr": float(p), "ssim": float(s_), "lpips": lp, "n": len(cams)}

    # ------------------------------------------------------------------ adaptive density control
    # FSGS' training loop (un-vendored) densifies with the published 3DGS clone / split / prune rules plus its own
    # proximity-guided unpooling; the published rules are restated here (Kerbl et al. 2023 section 5.2, UNPINNED: checked
    # against oracle/densify_oracle.py), the FSGS-specific unpooling is not (no source, no description of its constants
    # in /root/reference).  Everything runs on the device; the optimiser moments follow the Gaussians.
    _PARAM_ATTRS = ("_xyz", "_features", "_opacity", "_scaling", "_rotation")     # = order of the optimiser groups

    def cameras_extent(self) -> float:
        """Radius of the training cameras around their centroid x 1.1 (published `getNerfppNorm`)."""
        cams = self.scene.getTrainCameras()
        if not cams:
            return 1.0
        c = torch.stack([cam.camera_center.detach().float().cpu() for cam in cams])
        return float((c - c.mean(0, keepdim=True)).norm(dim=1).max() * 1.1) or 1.0

    def _swap_param(self, attr: str, tensor: torch.Tensor, moments=None):
        """Replace parameter `attr` by `tensor` in the model and in its optimiser group; `moments` maps the old
        (exp_avg, exp_avg_sq) to the new ones (None: start from zero, as a fresh parameter)."""
        g = self.gaussians
        old = getattr(g, attr)
        new = torch.nn.Parameter(tensor.contiguous())
        grp = self.optimizer.param_groups[self._PARAM_ATTRS.index(attr)]
        st = self.optimizer.state.pop(old, None)
        if st is not None and moments is not None:
            st["exp_avg"], st["exp_avg_sq"] = moments(st["exp_avg"]).contiguous(), moments(st["exp_avg_sq"]).contiguous()
            self.optimizer.state[new] = st
        grp["params"] = [new]
        setattr(g, attr, new)

    @torch.no_grad()
    def _append_gaussians(self, ext: dict, confidence: torch.Tensor):
        g = self.gaussians
        for attr in self._PARAM_ATTRS:
            e = ext[attr]
            self._swap_param(attr, torch.cat([getattr(g, attr).detach(), e]),
                             moments=lambda m, e=e: torch.cat([m, torch.zeros_like(e)]))
        g.confidence = torch.cat([g.confidence, confidence])
        g.xyz_gradient_accum = None           # statistics restart after every change of the set (published postfix)
        g.ensure_stats()

    @torch.no_grad()
    def _keep_gaussians(self, keep: torch.Tensor):
        g = self.gaussians
        g.ensure_stats()
        for attr in self._PARAM_ATTRS:
            self._swap_param(attr, getattr(g, attr).detach()[keep], moments=lambda m: m[keep])
        g.confidence = g.confidence[keep]
        g.xyz_gradient_accum, g.denom, g.max_radii2D = g.xyz_gradient_accum[keep], g.denom[keep], g.max_radii2D[keep]

    def _split_noise(self, n: int) -> torch.Tensor:
        """Standard-normal draws [n,3] for the split positions (seeded per trainer: runs are reproducible)."""
        dev = self.gaussians._xyz.device
        if getattr(self, "_noise_gen", None) is None:
            self._noise_gen = torch.Generator(device=dev)
            self._noise_gen.manual_seed(int(self.opt.seed) + 12345)
        return torch.randn((n, 3), generator=self._noise_gen, device=dev)

    @torch.no_grad()
    def densify_and_prune(self, max_grad: float, min_opacity: float, extent: float, max_screen_size: Optional[float]):
        """Published 3DGS `densify_and_prune`: clone small Gaussians with a large view-space gradient, split large ones
        into two (positions sampled from the Gaussian, scales / 1.6), then prune transparent, screen-filling and
        world-huge ones.  Returns (cloned, split, pruned)."""
        g, o = self.gaussians, self.opt
        g.ensure_stats()
        grads = g.xyz_gradient_accum / g.denom
        grads[grads.isnan()] = 0.0
        max_radii = g.max_radii2D.clone()
        # ---- clone
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (g.get_scaling.max(dim=1).values <= o.percent_dense * extent)
        n_clone = int(sel.sum())
        self._append_gaussians({a: getattr(g, a).detach()[sel] for a in self._PARAM_ATTRS}, g.confidence[sel])
        max_radii = torch.cat([max_radii, torch.zeros(n_clone, device=max_radii.device)])
        # ---- split (the clones carry a zero gradient)
        n_now = g._xyz.shape[0]
        padded = torch.zeros(n_now, device=g._xyz.device)
        padded[:grads.shape[0]] = grads.squeeze(-1)
        sel = (padded >= max_grad) & (g.get_scaling.max(dim=1).values > o.percent_dense * extent)
        n_split = int(sel.sum())
        N = 2
        stds = g.get_scaling[sel].repeat(N, 1)
        samples = self._split_noise(N * n_split) * stds
        rots = build_rotation(g._rotation.detach()[sel]).repeat(N, 1, 1)
        ext = {"_xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + g._xyz.detach()[sel].repeat(N, 1),
               "_scaling": torch.log(g.get_scaling[sel].repeat(N, 1) / (0.8 * N)),
               "_rotation": g._rotation.detach()[sel].repeat(N, 1),
               "_features": g._features.detach()[sel].repeat(N, 1, 1),
               "_opacity": g._opacity.detach()[sel].repeat(N)}
        self._append_gaussians(ext, g.confidence[sel].repeat(N))
        max_radii = torch.cat([max_radii, torch.zeros(N * n_split, device=max_radii.device)])
        keep = ~torch.cat([sel, torch.zeros(N * n_split, dtype=torch.bool, device=sel.device)])
        self._keep_gaussians(keep)
        max_radii = max_radii[keep]
        # ---- prune
        prune = g.get_opacity < min_opacity
        if max_screen_size:
            prune = prune | (max_radii > max_screen_size) | (g.get_scaling.max(dim=1).values > 0.1 * extent)
        n_prune = int(prune.sum())
        self._keep_gaussians(~prune)
        g.xyz_gradient_accum = None
        g.ensure_stats()
        return n_clone, n_split, n_prune

    @torch.no_grad()
    def reset_opacity(self):
        """Published `reset_opacity`: opacity <- min(opacity, 0.01), Adam moments of the opacity restart."""
        g = self.gaussians
        o = torch.min(g.get_opacity, torch.full_like(g.get_opacity, 0.01))
        self._swap_param("_opacity", torch.log(o / (1 - o)), moments=lambda m: torch.zeros_like(m))

    def _density_control(self, out: dict):
        """The per-iteration hook of the published training loop (after backward, before the optimiser step)."""
        g, o, it = self.gaussians, self.opt, self.iteration + 1
        if it >= o.densify_until_iter:
            return
        vis = out["visibility_filter"]
        vgrad = out["viewspace_grad"] if "viewspace_grad" in out else out["viewspace_points"].grad
        g.add_densification_stats(vgrad, vis, out["radii"])
        if it > o.densify_from_iter and it % o.densification_interval == 0:
            size = o.prune_screen_size if it > o.opacity_reset_interval else None
            self.densify_and_prune(o.densify_grad_threshold, o.prune_min_opacity, self.cameras_extent(), size)
        if it % o.opacity_reset_interval == 0:
            self.reset_opacity()

    def render_view(self, cam: Camera, scaling_modifier: float = 1.0):
        """-> {'render' [3,H,W], 'depth' [1,H,W], 'alpha' [1,H,W], ...} (diffusionGS.py:154-172)."""
        g = self.gaussians
        st = GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=self.background, scale_modifier=scaling_modifier,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=g.active_sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False)
        means2D = torch.zeros_like(g.get_xyz, requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(st)(g.get_xyz, means2D, g.get_opacity, shs=g.get_features,
                                                            scales=g.get_scaling, rotations=g.get_rotation,
                                                            confidence=g.confidence)
        return {"render": color, "depth": depth, "alpha": alpha, "viewspace_points": means2D,
                "visibility_filter": radii > 0, "radii": radii}

    def _pick_camera(self) -> Camera:
        pseudo = self.scene.getPseudoCameras()
        if pseudo and self._rng.random() < self.opt.pseudo_cam_sampling_rate:
            return pseudo[int(self._rng.integers(len(pseudo)))]
        cams = self.scene.getTrainCameras()
        return cams[int(self._rng.integers(len(cams)))]

    def _explicit_step(self, cam: Camera) -> Tuple[torch.Tensor, dict]:
        """One optimisation step WITHOUT autograd: raw parameters in, raw-parameter gradients out.  The activations and
        their chain rule run INSIDE the rasteriser's projection kernels (`syn3r_raster_preprocess_raw` / `_backward_raw`, round 6:
        `syn3r_gaussian_activate[_backward]`'s arithmetic, two launches and six intermediate tensors less per iteration; the
        autograd path keeps torch's three operators), the loss' value and gradient are `syn3r_photo_loss_step`'s two launches,
        and no screen-space `means2D` tensor is allocated per render.  Same arithmetic as the autograd path
        (`tests/test_trainer_gpu.py` holds one step of each against the oracle and against each other)."""
        from ..raster import rasterize_backward, rasterize_forward
        from .train_ops import l1_loss_step, photometric_loss_step
        g = self.gaussians
        # this step discards the rasteriser's confidence gradient: the per-Gaussian confidence is data here, as in the call sites
        # (model/diffusionGS.py:139,1640); a trainable one has to take the autograd path
        conf = g.confidence
        if conf is not None and getattr(conf, "requires_grad", False):
            raise ValueError("_explicit_step: a confidence tensor that requires grad needs train_step(explicit=False)")
        with torch.no_grad():
            st = GaussianRasterizationSettings(
                image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5), bg=self.background, scale_modifier=1.0,
                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=g.active_sh_degree,
                campos=cam.camera_center, prefiltered=False, debug=False)
            # log-scales / raw quaternions / logits in, THEIR gradients out (syn3r_raster_*_raw)
            color, radii, depth, alpha, rstate = rasterize_forward(g._xyz, g._features, g._opacity, g._scaling, g._rotation,
                                                                   g.confidence, st, raw_params=True)
            w = float(cam.cam_confidence)
            if self.opt.lambda_dssim > 0.0:
                loss, _, d_color = photometric_loss_step(color, cam.original_image, self.opt.lambda_dssim, w)
            else:
                loss, d_color = l1_loss_step(color, cam.original_image, w)
            d_m3, d_m2, d_sh, d_lg, d_ls, d_rr, _ = rasterize_backward(rstate, d_color)
            g._xyz.grad, g._features.grad, g._opacity.grad, g._scaling.grad, g._rotation.grad = d_m3, d_sh, d_lg.reshape(g._opacity.shape), d_ls, d_rr
        out = {"render": color, "depth": depth, "alpha": alpha, "viewspace_grad": d_m2, "visibility_filter": None,      # visible = radii > 0: add_densification_stats takes it from `radii` on the device
               "radii": radii}
        return loss, out

    def _one(self, dev) -> torch.Tensor:
        one = getattr(self, "_one_cache", None)
        if one is None or one.device != dev:
            one = self._one_cache = torch.ones((), dtype=torch.float32, device=dev)
        return one

    def train_step(self, cam: Optional[Camera] = None, explicit: Optional[bool] = None) -> torch.Tensor:
        """One optimisation step; returns the loss as a DEVICE scalar (no host synchronisation: the loop queues
        iterations back to back, `float(loss)` is the caller's choice).  `opt.use_lpips_loss` (set by the orchestrator
        around refine_GS, diffusionGS.py:1690,1697) adds `opt.lpips_weight` x LPIPS-VGG (`syn3r_amd.gs.lpips`, HIP) when the
        trainer has been given an `lpips` model with weights (the pretrained ones are not reachable offline: the caller
        loads them, as for CLIP / VAE / UNet); without one the switch is inert.
        `explicit` (default: whenever the LPIPS term is off): the step without autograd (`_explicit_step`)."""
        cam = cam or self._pick_camera()
        lpips_on = self.opt.use_lpips_loss and self.opt.lpips_weight > 0.0 and self.lpips is not None
        if explicit is None:
            explicit = not lpips_on
        if explicit:
            if lpips_on:
                raise ValueError("train_step(explicit=True) has no LPIPS term: the perceptual loss runs through autograd")
            loss, out = self._explicit_step(cam)
        else:
            out = self.render_view(cam)
            if self.opt.lambda_dssim > 0.0:
                loss = photometric_loss(out["render"], cam.original_image, self.opt.lambda_dssim, float(cam.cam_confidence))
            else:
                loss = l1_loss(out["render"], cam.original_image, weight=float(cam.cam_confidence))
            if lpips_on:
                # the perceptual term of the refine stage (diffusionGS.py:1690,1697; `--lpips_weight`): published LPIPS-VGG on the
                # whole render, weighted like the photometric term by the camera confidence.  How FSGS applies it is not visible
                # (un-vendored): UNPINNED.
                loss = loss + (self.opt.lpips_weight * float(cam.cam_confidence)) * self.lpips(out["render"].clamp(0, 1), cam.original_image)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
        changed = False
        if self.densify:
            n0 = self.gaussians._xyz.shape[0]
            self._density_control(out)
            changed = self.gaussians._xyz.shape[0] != n0
        if not changed:                       # (the gradients of this step belong to the old set of Gaussians)
            self.optimizer.step()
        self.iteration += 1
        return loss.detach()

    def _capacity_keys(self):
        """The (device, N, H, W) keys of the binning capacities the training cameras use; the camera list is walked once per
        list object and Gaussian count, not per iteration."""
        g = self.gaussians
        dev, n = g._xyz.device, g._xyz.shape[0]
        cams = self.scene.train_cameras[1.0]
        tag = (id(cams), len(cams), n)
        if getattr(self, "_ck_tag", None) != tag:
            from .. import raster
            shapes = sorted({(int(c.image_height), int(c.image_width)) for c in cams})
            self._ck_tag, self._ck_keys = tag, [raster.capacity_key(dev, n, h, w) for h, w in shapes]
        return self._ck_keys

    def _loop(self, first_iter: int, n: int) -> float:
        from .. import raster
        prev = raster.get_pair_count_mode()
        # async mode sizes the binning buffer from earlier renders of the same (N, H, W): one exact (sync) render per
        # camera first, so the capacity covers the view with the most (Gaussian, tile) pairs with 2x headroom
        raster.set_pair_count_mode("sync")
        if n > first_iter:
            with torch.no_grad():
                for cam in self.scene.train_cameras[1.0]:
                    self.render_view(cam)
        raster.set_pair_count_mode("async")          # no host round trip per render inside the loop
        self.iteration = first_iter                  # the density-control schedule counts from the start of this loop
        last = None

        def late_overflow(e):                        # an EARLIER render outgrew its pair capacity (reported late, see raster)
            if "async pair-count" not in str(e):
                raise e
            self.truncated_renders += int(getattr(e, "truncated", 1))

        try:
            for _ in range(first_iter, n):
                keys0 = self._capacity_keys() if self.densify else None
                cam = self._pick_camera()            # drawn ONCE per iteration: a retry renders the same view
                for attempt in range(4):
                    try:
                        last = self.train_step(cam)
                        break
                    except L.Syn3rError as e:        # raised by the render at the TOP of this step, about an earlier render
                        late_overflow(e)             # (whose step was applied): counted, reported in the scene record
                        if attempt == 3:             # (`truncated_renders`), the capacity raised - this iteration runs now;
                            raise                    # a retry may find ANOTHER pending truncated render: bounded
                if keys0 is not None:
                    keys1 = self._capacity_keys()
                    if keys1 != keys0:               # densification changed N: every (N, H, W) key is new.  Check what
                        try:                         # the old shapes still owe, then seed the new capacity from the old
                            raster.flush_pair_checks()           # one scaled by N_new / N_old (max over cameras is kept)
                        except L.Syn3rError as e:
                            late_overflow(e)
                        for k0, k1 in zip(keys0, keys1):
                            raster.carry_capacity(k0, k1, k1[1] / max(k0[1], 1))
            try:
                raster.flush_pair_checks()           # every remaining render's pair list was complete
            except L.Syn3rError as e:
                late_overflow(e)
        finally:
            raster.set_pair_count_mode(prev)
        return float(last) if last is not None else 0.0     # ONE synchronisation, at the end of the loop

    def training(self, first_iter: int = 0, epoch_indicator: int = 0, iterations: Optional[int] = None,
                 disable_densification: bool = False):
        """HOT LOOP A (diffusionGS.py:139): `opt.iterations` optimisation steps with the published adaptive density
        control (clone / split / prune every `densification_interval` iterations from `densify_from_iter` on, opacity
        reset every `opacity_reset_interval`), then the checkpoints the orchestrator's refine_GS looks for
        (`chkpnt{N}.pth` for the configured checkpoint iterations, else `chkpnt_latest.pth`, diffusionGS.py:1620-1624)
        when `scene.model_path` is set."""
        n = iterations if iterations is not None else self.opt.iterations
        self.densify = not disable_densification
        last = self._loop(first_iter, n)
        if self.scene.model_path:
            if n in self.checkpoint_iterations:
                self.save_checkpoint(n)
            self.save_checkpoint(n, latest=True)
        return last

    def finetune(self, first_iter: int = 0, refine_epoch: int = 0, disable_densification: bool = False,
                 pseudo_cam_sampling_rate: Optional[float] = None, iterations: Optional[int] = None):
        """diffusionGS.py:1640 — same loop (density control unless `disable_densification`, the flag refine_GS forwards,
        :1610), now also sampling the confidence-weighted pseudo-views; writes
        `refine_{epoch}_chkpnt{N}.pth` (the file the next cycle's refine_GS reloads, :1611-1618)."""
        if pseudo_cam_sampling_rate is not None:
            self.opt.pseudo_cam_sampling_rate = pseudo_cam_sampling_rate
        n = iterations if iterations is not None else self.opt.iterations
        self.densify = not disable_densification
        last = self._loop(first_iter, n)
        if self.scene.model_path:
            self.save_checkpoint(n, refine_epoch=refine_epoch)
        return last
