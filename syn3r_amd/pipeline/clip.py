"""CLIP image conditioning behind the pipeline's `image_encoder` slot, loaded from a LOCAL checkpoint directory.

The CLIP vision tower itself is out of this build's scope (two forward passes per `svd_render`, negligible
next to 200 UNet forwards): it runs on `transformers`' own `CLIPVisionModelWithProjection` when that package is importable,
and `from_pretrained` raises a clear error otherwise.  What IS restated here is the reference's preprocessing in front of it
(`model/SVD_2pass_prob_uncertain_post.py:229-258`): image to [-1, 1], Gaussian-prefiltered bicubic resize to 224 x 224
(`_resize_with_antialiasing`, `:107-135`: sigma = (factor - 1) / 2, kernel of ~4 sigma taps made odd, reflect padding,
`interpolate(bicubic, align_corners=True)`), back to [0, 1], CLIP mean / std normalisation (the `feature_extractor` call
with every other step switched off, `:249-256`).  Pinned by tests/golden/clip_preprocess.npz, the reference function's own
output (`oracle/gen_golden.py clip`).
"""
from __future__ import annotations

import json
from pathlib import Path
from types import SimpleNamespace
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as Fn

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _gauss_taps(n: int, sigma: float, dtype) -> torch.Tensor:
    x = torch.arange(n, dtype=dtype) - n // 2
    if n % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2.0 * sigma * sigma))
    return g / g.sum()


def resize_with_antialiasing(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """[N, C, H, W] -> [N, C, size]: separable Gaussian low-pass sized by the down-scaling factor, then bicubic."""
    h, w = x.shape[-2:]
    sig = (max((h / size[0] - 1.0) / 2.0, 0.001), max((w / size[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(4.0 * s, 3)) for s in sig]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    c = x.shape[1]
    kx = _gauss_taps(ks[1], sig[1], x.dtype).to(x.device)
    ky = _gauss_taps(ks[0], sig[0], x.dtype).to(x.device)
    y = Fn.pad(x, (ks[1] // 2, ks[1] - 1 - ks[1] // 2, 0, 0), mode="reflect")
    y = Fn.conv2d(y, kx.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)
    y = Fn.pad(y, (0, 0, ks[0] // 2, ks[0] - 1 - ks[0] // 2), mode="reflect")
    y = Fn.conv2d(y, ky.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return Fn.interpolate(y, size=size, mode="bicubic", align_corners=True)


def clip_pixel_values(image, mean: Sequence[float] = CLIP_MEAN, std: Sequence[float] = CLIP_STD,
                      value_range: Optional[float] = None) -> torch.Tensor:
    """The reference's `_encode_image` preprocessing of one image -> [1, 3, 224, 224] float32 CLIP input.
    PIL / uint8 HWC arrays are [0, 255] (VaeImageProcessor.pil_to_numpy); float HWC arrays and CHW / NCHW tensors are [0, 1]
    unless `value_range` says otherwise (255.0 for a float image in [0, 255]: the range is an argument, never guessed from
    the data - a dark [0, 255] image has no large value to guess from)."""
    if value_range is not None and value_range not in (1.0, 255.0):
        raise ValueError(f"value_range must be 1.0 or 255.0, got {value_range!r}")
    if isinstance(image, torch.Tensor):
        x = image.detach().float().cpu()
        if x.dim() == 3:
            x = x[None]
        if value_range == 255.0:
            x = x / 255.0
    else:
        a = np.asarray(image)
        if a.dtype == np.uint8:
            if value_range == 1.0:
                raise ValueError("a uint8 image is [0, 255]")
            a = a.astype(np.float32) / 255.0                      # VaeImageProcessor.pil_to_numpy
        else:
            a = a.astype(np.float32)
            if value_range == 255.0:
                a = a / 255.0
        x = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None]
    x = x * 2.0 - 1.0
    x = resize_with_antialiasing(x, (224, 224))
    x = (x + 1.0) / 2.0
    m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - m) / s


class ClipImageEncoder:
    """`image_encoder(image).image_embeds` [1, projection_dim] for the SVD pipelines."""

    def __init__(self, model, mean=CLIP_MEAN, std=CLIP_STD, device="cuda:0", dtype=torch.float16):
        self.model, self.mean, self.std = model, tuple(mean), tuple(std)
        self.device, self.dtype = torch.device(device), dtype

    @classmethod
    def from_pretrained(cls, image_encoder_dir, feature_extractor_dir=None, device="cuda:0", dtype=torch.float16):
        try:
            from transformers import CLIPVisionModelWithProjection
        except ImportError as e:          # pragma: no cover - transformers is part of this image
            raise RuntimeError("loading image_encoder/ needs the `transformers` package (CLIP is outside this build's HIP "
                               "scope); pass your own image_encoder module to StableVideoDiffusionPipeline instead") from e
        d = Path(image_encoder_dir)
        if not (d / "config.json").exists():
            raise FileNotFoundError(f"no CLIP vision checkpoint under {d} (config.json + *.safetensors expected)")
        model = CLIPVisionModelWithProjection.from_pretrained(str(d), local_files_only=True, torch_dtype=dtype)
        model = model.to(device).eval()
        mean, std = CLIP_MEAN, CLIP_STD
        if feature_extractor_dir is not None and (Path(feature_extractor_dir) / "preprocessor_config.json").exists():
            cfg = json.loads((Path(feature_extractor_dir) / "preprocessor_config.json").read_text())
            mean, std = cfg.get("image_mean", mean), cfg.get("image_std", std)
        return cls(model, mean, std, device, dtype)

    @torch.no_grad()
    def __call__(self, image):
        px = clip_pixel_values(image, self.mean, self.std).to(device=self.device, dtype=self.dtype)
        return SimpleNamespace(image_embeds=self.model(px).image_embeds)
