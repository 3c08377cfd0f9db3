"""Developer tool: GroupNorm / LayerNorm / temporal attention at the UNet's shapes (F = 14, B = 2): us and GB/s per launch."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
B, F = 2, 14


def dev_us(fn, n=10):
    fn(); torch.cuda.synchronize()
    with L.kernel_trace() as tr:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return {k: 1e3 * v[1] / n for k, v in tr.result.items()}


for (h, w, C) in ((72, 128, 320), (36, 64, 640), (18, 32, 1280), (9, 16, 1280), (72, 128, 640), (72, 128, 960), (36, 64, 1280), (36, 64, 1920)):
    M = B * F * h * w
    x = torch.randn(M, C, device=dev).half()
    g, b = torch.ones(C, device=dev).half(), torch.zeros(C, device=dev).half()
    nbytes = M * C * 2
    r = dev_us(lambda: ops.groupnorm(x, g, b, B * F, 1e-5, True))
    tot = sum(r.values())
    print(f"groupnorm 2D [{M},{C}]: " + ", ".join(f"{k} {v:.1f}" for k, v in r.items()) + f" | total {tot:.1f} us, {3 * nbytes / tot / 1e3:.0f} GB/s of 3 passes (min 2 passes: {2 * nbytes / tot / 1e3:.0f})")
    r = dev_us(lambda: ops.groupnorm(x, g, b, B, 1e-5, True))
    tot = sum(r.values())
    print(f"groupnorm 3D [{M},{C}]: total {tot:.1f} us, {3 * nbytes / tot / 1e3:.0f} GB/s")
    if C in (320, 640, 1280):
        r = dev_us(lambda: ops.layernorm(x, g, b))
        tot = sum(r.values())
        print(f"layernorm    [{M},{C}]: total {tot:.1f} us, {2 * nbytes / tot / 1e3:.0f} GB/s")
        qkv = torch.randn(M, 3 * C, device=dev).half()
        r = dev_us(lambda: ops.attention_temporal(qkv, B, F, h * w, C // 64))
        tot = sum(r.values())
        print(f"attn_temporal[{M},{C}]: total {tot:.1f} us, {4 * nbytes / tot / 1e3:.0f} GB/s")
