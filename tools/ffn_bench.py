"""Fused feed-forward (C = 320) against the two-kernel path at the UNet's level-0 shape (developer tool)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 258048
C, D = 320, 1280
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g, device=dev) * scale).half()
x, res = rnd(M, C), rnd(M, C)
w1, b1, w2, b2 = rnd(2 * D, C, scale=C ** -0.5), rnd(2 * D), rnd(C, D, scale=D ** -0.5), rnd(C)
wc, bc, _ = ops.pack_geglu_chunked(w1, b1)
wp, bp, _ = ops.pack_geglu(w1, b1)
flops = 2.0 * M * (2 * D * C + C * D)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t2 = timeit(lambda: ops.feedforward(x, wp, bp, D, w2, b2, residual=res))
t1 = timeit(lambda: ops.feedforward_fused(x, wc, bc, D, w2, b2, residual=res))
o1 = ops.feedforward_fused(x, wc, bc, D, w2, b2, residual=res).float()
o2 = ops.feedforward(x, wp, bp, D, w2, b2, residual=res).float()
print(f"M={M}: two kernels {t2:.3f} ms ({flops / t2 / 1e9:.0f} TFLOP/s)  fused {t1:.3f} ms ({flops / t1 / 1e9:.0f} TFLOP/s)  "
      f"max diff {float((o1 - o2).abs().max()):.3e} (scale {float(o2.abs().max()):.2f})")
