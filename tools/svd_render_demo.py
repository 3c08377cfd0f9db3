"""One complete `StableVideoDiffusionPipeline.__call__` at the reference's size (576x1024, F = 25, both passes) with the
full-size HIP UNet and HIP VAE (seeded weights) and a stand-in CLIP embedder — the call `DiffusionGS.svd_render`
makes (model/diffusionGS.py:1100), with fewer denoising steps.  Developer tool: end-to-end plumbing and wall time."""
import sys, time
from pathlib import Path
from types import SimpleNamespace
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import orchestrator as O
from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
from syn3r_amd.schedulers.scheduling_euler_discrete import EulerDiscreteScheduler, SVD_XT_SCHEDULER_CONFIG
from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
from syn3r_amd.vae import AutoencoderKLTemporalDecoder

variant = sys.argv[1] if len(sys.argv) > 1 else "post"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
F = int(sys.argv[3]) if len(sys.argv) > 3 else 25
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
unet = UNetSpatioTemporalConditionModel().init_random(dev, seed=0)
vae = AutoencoderKLTemporalDecoder(block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                                   layers_per_block=2, sample_size=768).init_random(dev, seed=1)


class Clip:                                   # stand-in for the CLIP vision tower (out of scope): image -> [1, 1024]
    def __call__(self, image):
        g = torch.Generator().manual_seed(int(np.asarray(image).sum()) % 1000)
        return SimpleNamespace(image_embeds=torch.randn(1, 1024, generator=g))


pipe = StableVideoDiffusionPipeline(vae, Clip(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG), variant=variant, device=dev)
print(f"models up in {time.perf_counter() - t0:.1f} s", flush=True)
rng = np.random.default_rng(0)
img = lambda: rng.random((576, 1024, 3), dtype=np.float32)
masks = torch.from_numpy((rng.random((F - 2, 72, 128)) > 0.5).astype(np.float32))
lam = O.search_hypers_v2(masks, diffusion_steps=steps) if F == 25 else torch.ones(steps, F, dtype=torch.float64)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = pipe([img()], temp_cond=[img() for _ in range(F - 1)], mask=masks, lambda_ts=lam, num_frames=F, decode_chunk_size=8,
           num_inference_steps=steps, output_type="pt")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
frames = out.frames if hasattr(out, "frames") else out
fr = frames[0] if isinstance(frames, (list, tuple)) else frames
print(f"variant {variant}, F = {F}, {steps} steps x 2 passes: {dt:.2f} s wall ({dt / steps:.2f} s per step incl. {F + 1} VAE encodes "
      f"and the {F}-frame decode); output {tuple(np.asarray(fr).shape) if not torch.is_tensor(fr) else tuple(fr.shape)}, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GB", flush=True)
