"""Synthetic workloads of SURVEY.md §8d (inputs only — no rasterisation or other compute here):
seeded Gaussian clouds and the FSGS-style camera matrices used by bench.py and the tests."""
from __future__ import annotations

import numpy as np
import torch


def look_at_camera(H, W, fovx_deg=60.0, znear=0.01, zfar=100.0, dtype=torch.float32, eye=(0.0, 0.0, 0.0)):
    """Camera at `eye` looking down +z (camera-0 frame of the synthetic scene).  Returns the
    FSGS-style transposed matrices (world_view_transform, full_proj_transform) and campos."""
    tanfovx = float(np.tan(np.deg2rad(fovx_deg) / 2))
    tanfovy = tanfovx * H / W
    w2c = np.eye(4)
    w2c[:3, 3] = -np.asarray(eye, dtype=np.float64)
    top, right = tanfovy * znear, tanfovx * znear
    P = np.zeros((4, 4))
    P[0, 0] = znear / right
    P[1, 1] = znear / top
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    view_t = torch.tensor(w2c.T, dtype=dtype)
    full_t = torch.tensor((P @ w2c).T, dtype=dtype)
    return view_t, full_t, torch.tensor(eye, dtype=dtype), tanfovx, tanfovy


def synthetic_gaussians(N, seed=1234, dtype=torch.float32, log_scale_mean=np.log(0.01), zrange=(2.0, 6.0), xy=1.0):
    """xyz U([-xy,xy]^2 x zrange), log-scale N(ln 0.01, 0.5), unit quats, opacity sigmoid(N(0,1.5)), SH deg 3 N(0,0.3)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(N, 3, generator=g, dtype=torch.float64)
    means = torch.stack([(u[:, 0] * 2 - 1) * xy, (u[:, 1] * 2 - 1) * xy, zrange[0] + u[:, 2] * (zrange[1] - zrange[0])], 1)
    scales = torch.exp(log_scale_mean + 0.5 * torch.randn(N, 3, generator=g, dtype=torch.float64))
    q = torch.randn(N, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(1.5 * torch.randn(N, generator=g, dtype=torch.float64))
    shs = 0.3 * torch.randn(N, 16, 3, generator=g, dtype=torch.float64)
    return means.to(dtype), scales.to(dtype), q.to(dtype), opac.to(dtype), shs.to(dtype)
