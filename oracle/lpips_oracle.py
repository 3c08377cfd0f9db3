"""TEST INFRASTRUCTURE (oracle) — LPIPS (VGG16) in plain torch fp32 with autograd: the checker for syn3r_amd/gs/lpips.py.

Restates the PUBLISHED `lpips.LPIPS(net='vgg', version='0.1')` (Zhang et al. 2018; the `lpips` package the un-vendored
FSGS trainer calls — neither is in /root/reference; call site of the switch: model/diffusionGS.py:1690,1697):
  x -> 2x - 1 -> (x - shift) / scale;  VGG16 features after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3;
  per layer: f / (||f||_2 over channels + 1e-10), squared difference, 1x1 conv with the `lin` weights, spatial mean; summed.
PARITY UNPINNED: no copy of the package or of its weights is reachable here; formula against formula on shared seeded weights.
Only tests/ import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as Fn

SHIFT = torch.tensor([-0.030, -0.088, -0.188]).view(1, 3, 1, 1)
SCALE = torch.tensor([0.458, 0.448, 0.450]).view(1, 3, 1, 1)
SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))


def lpips(pred: torch.Tensor, target: torch.Tensor, sd: dict, half_weights: bool = True) -> torch.Tensor:
    """pred, target [3,H,W] in [0,1] (fp32/fp64 CPU) -> scalar.  `half_weights`: round the weights to fp16 first, as the
    HIP path stores them."""
    q = (lambda t: t.half().to(pred.dtype)) if half_weights else (lambda t: t.to(pred.dtype))

    def features(x):
        x = ((2 * x[None] - 1) - SHIFT.to(x.dtype)) / SCALE.to(x.dtype)
        out = []
        for s, idxs in enumerate(SLICES):
            if s > 0:
                x = Fn.max_pool2d(x, 2, 2)
            for i in idxs:
                x = Fn.relu(Fn.conv2d(x, q(sd[f"net.slice{s + 1}.{i}.weight"]), q(sd[f"net.slice{s + 1}.{i}.bias"]), padding=1))
            out.append(x)
        return out

    fa, fb = features(pred), features(target)
    total = 0
    for k, (a, b) in enumerate(zip(fa, fb)):
        na = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        w = sd[f"lin{k}.model.1.weight"].to(pred.dtype)
        total = total + ((na - nb) ** 2 * w).sum(1).mean()
    return total
