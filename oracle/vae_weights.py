"""Reduced temporal-decoder VAE configuration and seeded inputs shared by oracle/gen_golden.py (which runs the
REFERENCE AutoencoderKLTemporalDecoder on the CPU in fp32) and tests/ (which run the HIP mirror).  Test
infrastructure: weights come from oracle.unet_weights.make_state_dict on the same name->shape table."""
from __future__ import annotations

import torch

SMALL_VAE_CONFIG = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D", "DownEncoderBlock2D"),
                        block_out_channels=(64, 128), layers_per_block=2, latent_channels=4, sample_size=32)

# four-level (x8) reduced VAE for the end-to-end pipeline case (the pipelines' latent grid is image / 8)
PIPELINE_VAE_CONFIG = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                           block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=4, sample_size=64)


def make_images(n=2, h=32, w=48, seed=0):
    g = torch.Generator().manual_seed(4000 + seed)
    return (torch.rand(n, 3, h, w, generator=g) * 2.0 - 1.0).to(torch.float16).float()


def make_latents(b=2, f=3, h=16, w=24, seed=0):
    g = torch.Generator().manual_seed(5000 + seed)
    return torch.randn(b * f, 4, h, w, generator=g).to(torch.float16).float()


# the SVD-XT checkpoint's VAE (config.json of stabilityai/stable-video-diffusion-img2vid-xt/vae, recalled from the public
# model card like the scheduler's): 97.7 M parameters
FULL_VAE_CONFIG = dict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                       block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, sample_size=768)
FULL_VAE_STRIDE = 8          # the fixture keeps every 8th pixel of the decoded frames


def make_full_image(h=576, w=1024, seed=0):
    """One smooth-plus-noise image in [-1, 1] at the pipelines' working size."""
    g = torch.Generator().manual_seed(4100 + seed)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    base = torch.stack([torch.sin(xs / 37 + c) * torch.cos(ys / 23 - c) for c in range(3)])
    return (0.7 * base + 0.3 * (torch.rand(3, h, w, generator=g) * 2 - 1)).clamp(-1, 1)[None].to(torch.float16).float()


def make_full_latents(f=2, h=72, w=128, seed=0):
    g = torch.Generator().manual_seed(5100 + seed)
    return torch.randn(f, 4, h, w, generator=g).to(torch.float16).float()
