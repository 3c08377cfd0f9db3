"""Forward (splat) and inverse (gather) depth warps on the HIP path.

Mirror of the reference's `solver_utils/forward_warp.py`: `forward_warp`,
`inverse_warp` keep their names, argument order, return structure and error
behaviour (bare asserts on shapes, forward_warp.py:162-168).  The kernels are
`syn3r_forward_warp` / `syn3r_inverse_warp` (include/syn3r_hip.h).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import _lib as L
from .consistency import _host

INVERSE_WARP_KEYS = ("warped_img", "warped_depth", "mask_warp", "mask_depth", "mask", "warped_masked_img",
                     "mask_inv", "mask_depth_strict", "warped_bg_mask", "mask_reproj", "soft_mask_reproj")


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise L.Syn3rError("forward_warp needs a HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def forward_warp(frame1: np.ndarray, mask1: Optional[np.ndarray], depth1: np.ndarray,
                 transformation1: np.ndarray, transformation2: np.ndarray, intrinsic1: np.ndarray,
                 intrinsic2: Optional[np.ndarray]):
    """Reference: solver_utils/forward_warp.py:141-182.

    numpy in, numpy out: (warped uint8 (h,w,3), mask bool (h,w), flow float64 (h,w,2)).
    """
    h, w = frame1.shape[:2]
    if intrinsic2 is None:
        intrinsic2 = np.copy(intrinsic1)
    assert frame1.shape == (h, w, 3)
    assert mask1 is None or mask1.shape == (h, w)
    assert depth1.shape == (h, w)
    assert transformation1.shape == (4, 4)
    assert transformation2.shape == (4, 4)
    assert intrinsic1.shape == (3, 3)
    assert intrinsic2.shape == (3, 3)
    dev = _device()
    lib = L.load()
    # host algebra as forward_warp.py:16,24
    T = np.matmul(np.asarray(transformation2, dtype=np.float64),
                  np.linalg.inv(np.asarray(transformation1, dtype=np.float64)))
    K1inv = np.linalg.inv(np.asarray(intrinsic1, dtype=np.float64))
    K2 = np.asarray(intrinsic2, dtype=np.float64)
    f = torch.from_numpy(np.ascontiguousarray(frame1, dtype=np.float64)).to(dev)
    d = torch.from_numpy(np.ascontiguousarray(depth1, dtype=np.float64)).to(dev)
    m = None
    if mask1 is not None:
        m = torch.from_numpy(np.ascontiguousarray(mask1).astype(np.uint8)).to(dev)
    warped = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    mask2 = torch.empty((h, w), dtype=torch.bool, device=dev)
    flow = torch.empty((h, w, 2), dtype=torch.float64, device=dev)
    need = lib.syn3r_forward_warp_workspace_bytes(h, w)
    ws = L.workspace(dev, need, "fw")
    rc = lib.syn3r_forward_warp(L.ptr(f), L.ptr(m), L.ptr(d), L.host_f64(T.flatten()), L.host_f64(K1inv.flatten()),
                                L.host_f64(K2.flatten()), h, w, L.ptr(warped), L.ptr(mask2), L.ptr(flow),
                                L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, "syn3r_forward_warp")
    return warped.cpu().numpy(), mask2.cpu().numpy(), flow.cpu().numpy()


def inverse_warp_batch(img, depth, depth_pseudo, pose1, pose2s, K, bandwidth=20, return_error=False):
    """`inverse_warp` for `nb` target views sharing one source view, in one ABI call.

    depth_pseudo: [nb, H, W]; pose2s: [nb, 4, 4].  Returns the reference's dict
    with a leading batch dimension on every tensor.
    """
    dev = L.require_gpu(img, depth, depth_pseudo)
    lib = L.load()
    img_c = img.detach().to(torch.float32).contiguous()
    dep_c = depth.detach().to(torch.float32).reshape(depth.shape[-2:]).contiguous()
    dp_c = depth_pseudo.detach().to(torch.float32).contiguous()
    nb, H, W = dp_c.shape
    if img_c.shape != (3, H, W) or dep_c.shape != (H, W):
        raise ValueError(f"shape mismatch: img {tuple(img_c.shape)}, depth {tuple(dep_c.shape)}, pseudo {tuple(dp_c.shape)}")
    p1 = _host(pose1)
    Kh = _host(K)
    Kinv = torch.inverse(Kh)
    p12, p21 = [], []
    for b in range(nb):
        p2 = _host(pose2s[b])
        p12.append(torch.matmul(p1, torch.inverse(p2)))   # forward_warp.py:217
        p21.append(p2 @ torch.inverse(p1))                # consistency.py:37 on the way back
    p12 = torch.stack(p12).flatten()
    p21 = torch.stack(p21).flatten()

    def new(shape, dtype):
        return torch.empty(shape, dtype=dtype, device=dev)

    out = {
        "warped_img": new((nb, 3, H, W), torch.float32),
        "warped_depth": new((nb, 1, H, W), torch.float32),
        "mask_warp": new((nb, H, W), torch.bool),
        "mask_depth": new((nb, H, W), torch.bool),
        "mask": new((nb, H, W), torch.bool),
        "warped_masked_img": new((nb, 3, H, W), torch.float32),
        "mask_inv": new((nb, H, W), torch.bool),
        "mask_depth_strict": new((nb, H, W), torch.bool),
        "warped_bg_mask": None,
        "mask_reproj": new((nb, H, W), torch.bool),
        "soft_mask_reproj": new((nb, H, W), torch.float32),
    }
    err = new((nb, H, W), torch.float32) if return_error else None
    need = lib.syn3r_inverse_warp_workspace_bytes(nb)
    ws = L.workspace(dev, need, "iw")
    rc = lib.syn3r_inverse_warp(
        L.ptr(img_c), L.ptr(dep_c), L.ptr(dp_c), L.host_f32(p12), L.host_f32(p21), L.host_f32(Kh.flatten()),
        L.host_f32(Kinv.flatten()), float(bandwidth), nb, H, W,
        L.ptr(out["warped_img"]), L.ptr(out["warped_depth"]), L.ptr(out["mask_warp"]), L.ptr(out["mask_depth"]),
        L.ptr(out["mask"]), L.ptr(out["warped_masked_img"]), L.ptr(out["mask_inv"]),
        L.ptr(out["mask_depth_strict"]), L.ptr(out["mask_reproj"]), L.ptr(out["soft_mask_reproj"]), L.ptr(err),
        L.ptr(ws), ws.numel(), L.stream_ptr(dev))
    L.check(rc, "syn3r_inverse_warp")
    if return_error:
        out["reproj_error"] = err
    return out


def inverse_warp(img, depth, depth_pseudo, pose1, pose2, K, bg_mask=None, bandwidth=20):
    """Reference: solver_utils/forward_warp.py:187-279.

    img [3,H,W], depth [1,H,W], depth_pseudo [1,H,W] on the GPU; poses 4x4 w2c; K 3x3.
    Returns the same 11-key dict.  `bg_mask` is None at every reference call
    site (diffusionGS.py:757,1339,1438); a non-None value is rejected.
    """
    if bg_mask is not None:
        raise NotImplementedError("bg_mask is unused by the SYN3R hot path and not supported")
    out = inverse_warp_batch(img, depth, depth_pseudo.reshape(1, *depth_pseudo.shape[-2:]), pose1,
                             pose2.reshape(1, 4, 4), K, bandwidth=bandwidth)
    res = {}
    for k in INVERSE_WARP_KEYS:
        v = out[k]
        res[k] = None if v is None else v[0]
    return res
