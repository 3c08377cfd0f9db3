"""Per-shape timing of the UNet contractions (developer tool; run on the GPU box)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
H = torch.float16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


F = int(sys.argv[1]) if len(sys.argv) > 1 else 14
BMS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
from syn3r_amd import _lib as L
BF = 2 * F
print(f"F={F}")
rows = []
for bm in BMS:
  L.load().syn3r_gemm_set_tile(bm)
  rows.append(("--", f"tile bm={bm}", 0, 0, 1, 0.0, 0.0))
  for name, hw, C in (("L0", 72 * 128, 320), ("L1", 36 * 64, 640), ("L2", 18 * 32, 1280), ("L3", 9 * 16, 1280)):
      M = BF * hw
      x = torch.randn(M, C, device=dev).to(H)
      for tag, N, K in (("lin CxC", C, C), ("qkv", 3 * C, C), ("ff1", 8 * C, C), ("ff2", C, 4 * C)):
          a = torch.randn(M, K, device=dev).to(H)
          w = (torch.randn(N, K, device=dev) * K ** -0.5).to(H)
          ms = timeit(lambda: ops.linear(a, w))
          rows.append((name, tag, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
      a = torch.randn(M, C, device=dev).to(H)
      w8 = (torch.randn(8 * C, C, device=dev) * C ** -0.5).to(H)
      wp, bp, D = ops.pack_geglu(w8, torch.zeros(8 * C, device=dev, dtype=H))
      ms = timeit(lambda: ops.linear_geglu(a, wp, bp, D))
      rows.append((name, "ff1+geglu", M, 8 * C, C, ms, 2.0 * M * 8 * C * C / ms / 1e9))
      h, w_ = {"L0": (72, 128), "L1": (36, 64), "L2": (18, 32), "L3": (9, 16)}[name]
      xi = x.view(BF, h, w_, C)
      wc = (torch.randn(C, 3, 3, C, device=dev) * (9 * C) ** -0.5).to(H)
      ms = timeit(lambda: ops.conv3x3(xi, wc))
      rows.append((name, "conv3x3", M, C, 9 * C, ms, 2.0 * M * C * 9 * C / ms / 1e9))
      wt = (torch.randn(C, 3, C, device=dev) * (3 * C) ** -0.5).to(H)
      ms = timeit(lambda: ops.tconv3(x, wt, None, 2, F, hw))
      rows.append((name, "tconv3", M, C, 3 * C, ms, 2.0 * M * C * 3 * C / ms / 1e9))
      heads = C // 64
      qkv = torch.randn(M, 3 * C, device=dev).to(H)
      ms = timeit(lambda: ops.attention(qkv, BF, hw, heads), n=3)
      rows.append((name, "attn_sp", M, hw, 64, ms, 4.0 * BF * heads * hw * hw * 64 / ms / 1e9))
      ms = timeit(lambda: ops.attention_temporal(qkv, 2, F, hw, heads))
      rows.append((name, "attn_t", M, F, 64, ms, 4.0 * 2 * hw * heads * F * F * 64 / ms / 1e9))
      g = torch.ones(C, device=dev, dtype=H)
      ms = timeit(lambda: ops.groupnorm(x, g, g, BF, 1e-5, True))
      rows.append((name, "gn+silu", M, C, 0, ms, 4.0 * M * C / ms / 1e6))      # GB/s (read twice + write once -> 3 passes; report 2-pass bytes)
      ms = timeit(lambda: ops.layernorm(x, g, g))
      rows.append((name, "layernorm", M, C, 0, ms, 4.0 * M * C / ms / 1e6))
      x8 = torch.randn(M, 8 * C, device=dev).to(H)
      ms = timeit(lambda: ops.geglu(x8))
      rows.append((name, "geglu", M, 4 * C, 0, ms, 2.0 * M * 12 * C / ms / 1e6))
for r in rows:
    unit = "TFLOP/s" if r[4] else "GB/s"
    print(f"{r[0]:3s} {r[1]:10s} M={r[2]:7d} N={r[3]:6d} K={r[4]:6d}  {r[5]:8.3f} ms  {r[6]:9.1f} {unit}")
