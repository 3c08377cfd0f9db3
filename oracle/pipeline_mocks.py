"""Deterministic stand-ins for the out-of-scope modules of the SVD pipelines (CLIP, VAE) and a cheap
closed-form "UNet", used to pin the PIPELINE LOOP LOGIC (tile slicing and stitching, time flips,
CFG, forward/backward blend, scheduler wiring) against the reference's own `__call__`
(oracle/gen_golden.py runs the reference pipeline class with these mocks on the CPU).
Test infrastructure only."""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch


class MockImageEncoder(torch.nn.Module):
    """image -> .image_embeds [1, 1024]: fixed sinusoidal features of the image's channel means."""

    def __init__(self, dtype=torch.float32):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1, dtype=dtype), requires_grad=False)

    def forward(self, image):
        if not isinstance(image, torch.Tensor):      # HWC numpy image, as the orchestrator passes it
            import numpy as np
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).permute(2, 0, 1)
        x = image.float()
        if x.dim() == 3:
            x = x[None]
        m = x.mean(dim=(0, 2, 3)).to("cpu")                                  # [3]
        k = torch.arange(1024, dtype=torch.float32)
        emb = torch.sin(0.01 * k * (1.0 + m[0])) + torch.cos(0.02 * k * (1.0 + m[1])) * m[2]
        return SimpleNamespace(image_embeds=emb[None].to(device=image.device, dtype=self.anchor.dtype))


class MockVAE:
    """encode(x).latent_dist.mode(): 8x8 average pooling + a fixed 3->4 channel mix."""
    dtype = torch.float32
    config = SimpleNamespace(force_upcast=False, scaling_factor=0.18215, block_out_channels=(1, 1, 1, 1))
    MIX = torch.tensor([[0.9, -0.3, 0.2], [0.1, 0.8, -0.4], [-0.5, 0.2, 0.7], [0.3, 0.3, 0.3]])

    def to(self, *a, **k):
        return self

    def decode(self, z, num_frames=None):
        """latents -> images: pseudo-inverse of the channel mix, 8x nearest upsampling."""
        inv = torch.linalg.pinv(self.MIX.to(z.device).float())
        img = torch.einsum("co,bohw->bchw", inv, z.float())
        return SimpleNamespace(sample=torch.nn.functional.interpolate(img, scale_factor=8, mode="nearest"))

    def encode(self, image):
        pooled = torch.nn.functional.avg_pool2d(image.float(), 8)
        lat = torch.einsum("oc,bchw->bohw", self.MIX.to(image.device), pooled)
        return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: lat))


class MockUNet(torch.nn.Module):
    """y[b,f] = W x[b,f] + 0.1 x[b,f+1,:4] + 0.05 sin(t) + 0.01 mean(ehs[b]) + 0.001 added[b,0], tanh-squashed.
    Elementwise / 1x1 only, so tiles of the latent grid commute with it exactly as with the real UNet's
    interface; signature and `config` fields as the pipelines use them."""

    def __init__(self, dtype=torch.float32):
        super().__init__()
        g = torch.Generator().manual_seed(99)
        self.W = torch.nn.Parameter((torch.randn(4, 8, generator=g) * 0.4).to(dtype), requires_grad=False)
        self.config = SimpleNamespace(in_channels=8, addition_time_embed_dim=256, num_frames=25, sample_size=96)
        self.add_embedding = SimpleNamespace(linear_1=SimpleNamespace(in_features=768))

    def forward(self, sample, timestep, encoder_hidden_states=None, added_time_ids=None, return_dict=False):
        x = sample
        W = self.W.to(device=x.device, dtype=x.dtype)
        y = torch.einsum("oc,bfchw->bfohw", W, x)
        y = y + 0.1 * torch.roll(x[:, :, :4], shifts=-1, dims=1)
        t = float(timestep)
        bias = 0.05 * math.sin(t) + 0.01 * encoder_hidden_states.float().mean(dim=(1, 2)) + 0.001 * added_time_ids[:, 0].float()
        y = torch.tanh(y + bias.to(x.dtype)[:, None, None, None, None])
        return (y,)


def pipeline_inputs(seed=0, F=25, H=576, W=1024):
    g = torch.Generator().manual_seed(seed)
    image = [torch.rand(3, H, W, generator=g)]
    temp_cond = [torch.rand(3, H, W, generator=g) for _ in range(F - 1)]
    mask = torch.rand(F - 2, H // 8, W // 8, generator=g)
    lam = (torch.rand(100, F, generator=g) > 0.4).double()
    lam[:, 0] = 1.0
    lam[:, -1] = 1.0
    latents = torch.randn(1, F, 4, H // 8, W // 8, generator=g)
    noise = torch.randn(1, 3, H, W, generator=g)
    return dict(image=image, temp_cond=temp_cond, mask=mask, lambda_ts=lam, latents=latents, noise=noise)
