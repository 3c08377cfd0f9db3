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
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..raster import GaussianRasterizationSettings, GaussianRasterizer
from .train_ops import FusedAdam, l1_loss, photometric_loss


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


class _Scene:
    def __init__(self, cams): self._train = list(cams)
    def getTrainCameras(self): return self._train


class GSTrainer:
    def __init__(self, gaussians: GaussianModel, train_cameras: Sequence[Camera], opt: Optional[OptimizationParams] = None,
                 background=(0.0, 0.0, 0.0)):
        self.gaussians, self.opt = gaussians, opt or OptimizationParams()
        self.scene = _Scene(train_cameras)
        self.pseudo_cameras: List[Camera] = []
        self.dust3r = None
        self.checkpoint_iterations: List[int] = []
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

    def update_cameras(self, views, poses, K, cam_confidences, append: bool = True):
        """diffusionGS.py:1631 — register SVD pseudo-views ([3,H,W] tensors + w2c poses) with their confidence."""
        if np.isscalar(cam_confidences):
            cam_confidences = [float(cam_confidences)] * len(views)
        cams = [Camera.from_w2c(np.asarray(p), np.asarray(K), v.shape[1], v.shape[2], image=v, cam_confidence=c,
                                data_device=self.gaussians._xyz.device) for v, p, c in zip(views, poses, cam_confidences)]
        self.pseudo_cameras = (self.pseudo_cameras + cams) if append else cams

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
        if self.pseudo_cameras and self._rng.random() < self.opt.pseudo_cam_sampling_rate:
            return self.pseudo_cameras[int(self._rng.integers(len(self.pseudo_cameras)))]
        cams = self.scene.getTrainCameras()
        return cams[int(self._rng.integers(len(cams)))]

    def train_step(self, cam: Optional[Camera] = None) -> float:
        cam = cam or self._pick_camera()
        out = self.render_view(cam)
        if self.opt.lambda_dssim > 0.0:
            loss = photometric_loss(out["render"], cam.original_image, self.opt.lambda_dssim, float(cam.cam_confidence))
        else:
            loss = l1_loss(out["render"], cam.original_image, weight=float(cam.cam_confidence))
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return float(loss.detach())

    def training(self, first_iter: int = 0, epoch_indicator: int = 0, iterations: Optional[int] = None):
        """HOT LOOP A (diffusionGS.py:139): `opt.iterations` optimisation steps."""
        n = iterations if iterations is not None else self.opt.iterations
        last = 0.0
        for _ in range(first_iter, n):
            last = self.train_step()
        return last

    def finetune(self, first_iter: int = 0, refine_epoch: int = 0, disable_densification: bool = True,
                 pseudo_cam_sampling_rate: Optional[float] = None, iterations: Optional[int] = None):
        """diffusionGS.py:1640 — same loop, now also sampling the confidence-weighted pseudo-views."""
        if pseudo_cam_sampling_rate is not None:
            self.opt.pseudo_cam_sampling_rate = pseudo_cam_sampling_rate
        return self.training(first_iter, refine_epoch, iterations)
