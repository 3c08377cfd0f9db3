"""Deterministic synthetic UNet weights keyed by parameter name (test infrastructure).

Both oracle/gen_golden.py (which loads them into the REFERENCE UNet) and tests/ (which load them
into the HIP UNet) call `make_state_dict` with the same name->shape table, so the fixture only
stores the reference's output."""
from __future__ import annotations

import math
import zlib

import torch

SMALL_CONFIG = dict(in_channels=8, out_channels=4, block_out_channels=(64, 128, 128, 128),
                    num_attention_heads=(1, 2, 2, 2), cross_attention_dim=64, addition_time_embed_dim=64,
                    projection_class_embeddings_input_dim=192, layers_per_block=1, num_frames=5)

# the same reduced UNet inside the SVD pipelines: their CLIP embedding is 1024 wide and they run 25 frames
PIPELINE_CONFIG = dict(SMALL_CONFIG, cross_attention_dim=1024, num_frames=25)


def make_state_dict(shapes, seed=0):
    sd = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
        if name.endswith("mix_factor"):
            t = torch.randn(shape, generator=g)
        elif ("norm" in name.split(".")[-2]) and name.endswith(".weight"):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        sd[name] = t.to(torch.float16).to(torch.float32)   # exactly representable in fp16 on both sides
    return sd


def make_inputs(B, F, h, w, seed=0, cross=64):
    g = torch.Generator().manual_seed(1000 + seed)
    sample = torch.randn(B, F, 8, h, w, generator=g).to(torch.float16).float()
    ehs = torch.randn(B, 1, cross, generator=g).to(torch.float16).float()
    added = torch.tensor([[6.0, 127.0, 0.02]] * B)
    timestep = torch.tensor(1.6378)
    return sample, timestep, ehs, added
