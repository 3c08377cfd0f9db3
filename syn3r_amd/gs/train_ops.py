"""Fused trainer-loop operators on the HIP library: the photometric L1 loss and the Adam
update that `gsTrainer.training()/finetune()` (call sites `model/diffusionGS.py:139,1640`; FSGS submodule not
vendored) run as torch elementwise chains.  No CPU fallback: they raise `Syn3rError` without the extension."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from .. import _lib as L


def _l1_forward(image: torch.Tensor, target: torch.Tensor, weight: float):
    L.require_gpu(image, target)
    if image.shape != target.shape or image.dtype != torch.float32 or target.dtype != torch.float32:
        raise ValueError("l1_loss: image and target must be float32 tensors of the same shape")
    image, target = image.contiguous(), target.contiguous()
    lib = L.load()
    n = image.numel()
    loss = torch.empty((), dtype=torch.float32, device=image.device)
    ws = L.workspace(image.device, lib.syn3r_l1_loss_workspace_bytes(n), "l1")
    L.check(lib.syn3r_l1_loss(L.ptr(image), L.ptr(target), n, float(weight), L.ptr(loss), L.ptr(ws), ws.numel(),
                              L.stream_ptr(image.device)), "l1_loss")
    return loss, image, target


def _l1_backward(image: torch.Tensor, target: torch.Tensor, weight: float, grad_loss):
    go = grad_loss.to(torch.float32).contiguous() if grad_loss is not None else None      # device scalar; None = 1
    grad = torch.empty_like(image)
    L.check(L.load().syn3r_l1_loss_backward(L.ptr(image), L.ptr(target), image.numel(), float(weight), L.ptr(go),
                                            L.ptr(grad), L.stream_ptr(image.device)), "l1_loss_backward")
    return grad


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image: torch.Tensor, target: torch.Tensor, weight: float):
        loss, image, target = _l1_forward(image, target, weight)
        ctx.save_for_backward(image, target)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor):
        L.join_active_trace()
        image, target = ctx.saved_tensors
        return _l1_backward(image, target, ctx.weight, grad_loss), None, None


def l1_loss_step(image: torch.Tensor, target: torch.Tensor, weight: float = 1.0, grad_loss: torch.Tensor = None):
    """Value and image gradient of `l1_loss` without autograd (the two launches `_L1Loss` makes): (loss, grad_image)."""
    loss, image, target = _l1_forward(image.detach(), target.detach(), weight)
    return loss, _l1_backward(image, target, weight, grad_loss)


def l1_loss(image: torch.Tensor, target: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """`weight * (image - target).abs().mean()` as one read of both images (forward) and one read + one write
    (backward); the upstream gradient stays on the device."""
    return _L1Loss.apply(image, target, weight)


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image: torch.Tensor, target: torch.Tensor, lambda_dssim: float, weight: float):
        L.require_gpu(image, target)
        if image.shape != target.shape or image.dim() != 3 or image.dtype != torch.float32 or target.dtype != torch.float32:
            raise ValueError("photometric_loss: image and target must be float32 [C,H,W] tensors of the same shape")
        image, target = image.contiguous(), target.contiguous()
        lib = L.load()
        C_, H_, W_ = image.shape
        # own buffer, not the shared workspace cache: the derivative maps must survive until backward
        ws = torch.empty(lib.syn3r_photo_loss_workspace_bytes(C_, H_, W_), dtype=torch.uint8, device=image.device)
        out = torch.empty(3, dtype=torch.float32, device=image.device)
        L.check(lib.syn3r_photo_loss(L.ptr(image), L.ptr(target), C_, H_, W_, float(lambda_dssim), float(weight),
                                     L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr(image.device)), "photo_loss")
        ctx.save_for_backward(image, target, ws)
        ctx.args = (float(lambda_dssim), float(weight))
        ctx.mark_non_differentiable(out)
        ctx.parts = out
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor, _grad_parts):
        L.join_active_trace()
        image, target, ws = ctx.saved_tensors
        lam, weight = ctx.args
        C_, H_, W_ = image.shape
        go = grad_loss.to(torch.float32).contiguous()
        grad = torch.empty_like(image)
        L.check(L.load().syn3r_photo_loss_backward(L.ptr(image), L.ptr(target), C_, H_, W_, lam, weight, L.ptr(go),
                                                   L.ptr(ws), L.ptr(grad), L.stream_ptr(image.device)),
                "photo_loss_backward")
        return grad, None, None, None


def photometric_loss(image: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2, weight: float = 1.0,
                     return_parts: bool = False):
    """The published 3DGS loss `weight * ((1 - lambda) * L1 + lambda * (1 - SSIM))` on [C,H,W] images, fused: one
    tile pass forward (also storing the SSIM derivative maps), one tile pass backward.  `return_parts`: also the
    device tensor [loss, L1, SSIM]."""
    loss, parts = _PhotoLoss.apply(image, target, lambda_dssim, weight)
    return (loss, parts) if return_parts else loss


def photometric_loss_step(image: torch.Tensor, target: torch.Tensor, lambda_dssim: float = 0.2, weight: float = 1.0,
                          grad_loss: torch.Tensor = None):
    """Value AND image gradient of `photometric_loss` without autograd (`syn3r_photo_loss_step`: the two tile passes, the
    scalar sums formed by the gradient pass' first block instead of a launch of their own).  Returns (loss - a view of
    parts[0] -, parts [loss, L1, SSIM], grad_image); `grad_loss`: device scalar, default 1.  Same bits as
    `_PhotoLoss.forward` + `.backward`."""
    dev = L.require_gpu(image, target)
    if image.shape != target.shape or image.dim() != 3 or image.dtype != torch.float32 or target.dtype != torch.float32:
        raise ValueError("photometric_loss_step: image and target must be float32 [C,H,W] tensors of the same shape")
    image, target = image.detach().contiguous(), target.detach().contiguous()
    lib = L.load()
    C_, H_, W_ = image.shape
    ws = L.workspace(dev, lib.syn3r_photo_loss_workspace_bytes(C_, H_, W_), "photo_step")     # the maps die with the call
    parts = torch.empty(3, dtype=torch.float32, device=dev)
    grad = torch.empty_like(image)
    go = grad_loss.to(torch.float32).contiguous() if grad_loss is not None else None
    L.check(lib.syn3r_photo_loss_step(L.ptr(image), L.ptr(target), C_, H_, W_, float(lambda_dssim), float(weight), L.ptr(go),
                                      L.ptr(parts), L.ptr(grad), L.ptr(ws), ws.numel(), L.stream_ptr(dev)), "photo_loss_step")
    return parts[0], parts, grad


def image_metrics(image: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Device tensor [PSNR (dB, peak 1), SSIM] of two float32 [C,H,W] images in [0,1]: the MSE from `syn3r_image_mse`
    (deterministic two-level sum) and the SSIM the fused photometric-loss kernel computes (published 3DGS window).
    No host synchronisation."""
    L.require_gpu(image, target)
    image, target = image.detach().contiguous(), target.detach().contiguous()
    if image.shape != target.shape or image.dim() != 3 or image.dtype != torch.float32:
        raise ValueError("image_metrics: float32 [C,H,W] images of the same shape")
    lib = L.load()
    n = image.numel()
    mse = torch.empty((), dtype=torch.float32, device=image.device)
    ws = L.workspace(image.device, lib.syn3r_l1_loss_workspace_bytes(n), "l1")
    L.check(lib.syn3r_image_mse(L.ptr(image), L.ptr(target), n, L.ptr(mse), L.ptr(ws), ws.numel(),
                                L.stream_ptr(image.device)), "image_mse")
    _, parts = _PhotoLoss.apply(image, target, 1.0, 1.0)
    psnr = -10.0 * torch.log10(mse.clamp_min(1e-12))
    return torch.stack([psnr, parts[2]])


def knn3_mean_dist2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point of a [n,3] fp32 cloud (n >= 4) to its three nearest neighbours:
    `distCUDA2` of the simple-knn extension, which FSGS' create_from_pcd turns into the initial Gaussian scales
    (model/diffusionGS.py:1685-1687 -> reset_gaussians_from_pcd).  Exact search on the device (csrc/knn.hip)."""
    L.require_gpu(points)
    if points.dim() != 2 or points.shape[1] != 3 or points.dtype != torch.float32:
        raise ValueError("knn3_mean_dist2: points must be a float32 [n,3] tensor")
    pts = points.detach().contiguous()
    n = pts.shape[0]
    lib = L.load()
    out = torch.empty(n, dtype=torch.float32, device=pts.device)
    ws = L.workspace(pts.device, lib.syn3r_knn3_workspace_bytes(n), "knn3")
    L.check(lib.syn3r_knn3_mean_dist2(L.ptr(pts), n, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr(pts.device)), "knn3_mean_dist2")
    return out


class FusedAdam:
    """`torch.optim.Adam(param_groups, eps=...)` (no weight decay / amsgrad) with one kernel per parameter tensor.
    Keeps torch's `param_groups` / `state` layout so checkpoints and lr schedules written for the torch optimiser
    keep working."""

    def __init__(self, param_groups: Iterable[dict], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.param_groups: List[dict] = []
        for g in param_groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        self.state: dict = {}

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self):
        """One update of every parameter that has a gradient: ONE launch per (beta1, beta2) and up to 8 tensors
        (`syn3r_adam_step_multi`; the trainer's five / six groups share their betas), element for element `torch.optim.Adam`."""
        import ctypes as C
        lib = L.load()
        batches: dict = {}
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                L.require_gpu(p)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("FusedAdam: parameters must be contiguous float32")
                st = self.state.get(p)
                if st is None:
                    st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                st["step"] += 1
                grad = p.grad.contiguous()
                batches.setdefault((float(b1), float(b2), p.device), []).append((p, grad, st, float(g["lr"]), float(g["eps"])))
        for (b1, b2, dev), items in batches.items():
            for k0 in range(0, len(items), 8):
                chunk = items[k0:k0 + 8]
                n = len(chunk)
                ptrs = lambda sel: (C.c_void_p * n)(*[sel(it) for it in chunk])
                rc = lib.syn3r_adam_step_multi(
                    n, ptrs(lambda it: it[0].data_ptr()), ptrs(lambda it: it[1].data_ptr()), ptrs(lambda it: it[2]["exp_avg"].data_ptr()),
                    ptrs(lambda it: it[2]["exp_avg_sq"].data_ptr()), (C.c_longlong * n)(*[it[0].numel() for it in chunk]),
                    (C.c_float * n)(*[it[3] for it in chunk]), b1, b2, (C.c_float * n)(*[it[4] for it in chunk]),
                    (C.c_int * n)(*[int(it[2]["step"]) for it in chunk]), L.stream_ptr(dev))
                L.check(rc, "adam_step_multi")
