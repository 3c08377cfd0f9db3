"""Full-size SVD VAE (97.7 M parameters, seeded random weights) on the HIP operators: encode of 576x1024 images and
decode of 8-frame chunks, as `svd_render` uses them (26 encodes + 25 decoded frames per call).  Developer tool."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
from syn3r_amd.unet import ops
from syn3r_amd.vae import AutoencoderKLTemporalDecoder

dev = torch.device("cuda", 0)
m = AutoencoderKLTemporalDecoder(block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                                 layers_per_block=2, sample_size=768)
m.init_random(dev, seed=7)
Hh, Ww = 576, 1024
g = torch.Generator().manual_seed(0)
img = (torch.rand(1, 3, Hh, Ww, generator=g) * 2 - 1).to(dev)
z = torch.randn(8, 4, Hh // 8, Ww // 8, generator=g).to(dev)


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    ops.FLOPS.update(enabled=True, gemm=0.0, attn=0.0)
    fn(); torch.cuda.synchronize()
    ops.FLOPS["enabled"] = False
    fl = ops.FLOPS["gemm"]
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, fl


ms, fl = timed(lambda: m.encode(img))
print(f"encode 1 x 576x1024: {ms:.1f} ms, {fl / 1e12:.2f} TFLOP in contractions -> {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
ms, fl = timed(lambda: m.decode(z, num_frames=8))
print(f"decode 8 frames 576x1024: {ms:.1f} ms, {fl / 1e12:.2f} TFLOP -> {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
y = m.decode(z, num_frames=8).sample
print("decoded", tuple(y.shape), "finite", bool(torch.isfinite(y).all()), "peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
with L.kernel_trace() as tr:
    m.decode(z, num_frames=8); torch.cuda.synchronize()
tot = sum(v[1] for v in tr.result.values())
for k, (c, t) in sorted(tr.result.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:28s} {c:5d} launches {t:9.2f} ms {100 * t / tot:5.1f} %")
