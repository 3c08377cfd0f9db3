"""Reprojection consistency on the HIP path.

Mirror of the reference's `solver_utils/consistency.py` (same function names,
argument order and meaning); the per-pixel work runs in
`syn3r_reproj_error` (include/syn3r_hip.h), the 4x4 / 3x3 host algebra stays in
torch exactly as the reference writes it (consistency.py:21,37).
"""
from __future__ import annotations

import torch

from .. import _lib as L


def _host(m: torch.Tensor) -> torch.Tensor:
    return m.detach().to(device="cpu", dtype=torch.float32)


def relative_poses(pose1: torch.Tensor, pose2: torch.Tensor):
    """(pose2 @ inv(pose1), pose1 @ inv(pose2)) as the reference forms them
    (consistency.py:37 `pose2@torch.inverse(pose1)`)."""
    p1, p2 = _host(pose1), _host(pose2)
    return p2 @ torch.inverse(p1), p1 @ torch.inverse(p2)


def consistency_check_with_depth(depth1, pose1, intrinsics1, depth2, pose2, intrinsics2):
    """Reference: solver_utils/consistency.py:44-91.

    depth1, depth2: (h, w) float tensors on the GPU; poses 4x4 w2c; intrinsics 3x3.
    Returns the (h, w) reprojection error in pixels.
    """
    if depth1.dim() != 2 or depth2.shape != depth1.shape:
        raise ValueError(f"depth maps must both be (h, w); got {tuple(depth1.shape)} and {tuple(depth2.shape)}")
    dev = L.require_gpu(depth1, depth2)
    lib = L.load()
    d1 = depth1.detach().to(torch.float32).contiguous()
    d2 = depth2.detach().to(torch.float32).contiguous()
    h, w = d1.shape
    T12, T21 = relative_poses(pose1, pose2)
    K1 = _host(intrinsics1)
    K2 = _host(intrinsics2)
    K1inv = torch.inverse(K1)
    err = torch.empty_like(d1)
    rc = lib.syn3r_reproj_error(L.ptr(d1), L.ptr(d2), L.host_f32(T12.flatten()), L.host_f32(T21.flatten()),
                                L.host_f32(K1.flatten()), L.host_f32(K1inv.flatten()), L.host_f32(K2.flatten()),
                                h, w, L.ptr(err), L.stream_ptr(dev))
    L.check(rc, "syn3r_reproj_error")
    return err
