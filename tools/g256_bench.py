"""The gated projection on the 256 x 256 tile (k_gemm_g256) against the 256 x 320 kernel (k_gemm_z<0>), isolated, interleaved on
one box (developer tool).  Per-launch device time from the library's kernel trace."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401
from syn3r_amd import _lib as L
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
H = torch.float16
g = torch.Generator(device="cpu").manual_seed(1)
for M, C, D in ((64512, 640, 2560), (16128, 1280, 5120)):
    x = torch.randn(M, C, generator=g).to(H).to(dev)
    w1 = (torch.randn(2 * D, C, generator=g) * C ** -0.5).to(H).to(dev)
    b1 = torch.randn(2 * D, generator=g).to(H).to(dev)
    w2 = (torch.randn(C, D, generator=g) * D ** -0.5).to(H).to(dev)
    res = torch.randn(M, C, generator=g).to(H).to(dev)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    w64, b64, _ = ops.pack_geglu64(w1, b1)
    a = ops.feedforward(x, wp, bp, D, w2, None, residual=res)
    b = ops.feedforward(x, wp, bp, D, w2, None, residual=res, packed64=(w64, b64))
    torch.cuda.synchronize()
    print(f"[{M},{2 * D},{C}] equal: {torch.equal(a, b)}")
    for rep in range(3):
        for name, pk in (("z<0> 256x320", None), ("g256 256x256", (w64, b64))):
            with L.kernel_trace(detail=True) as tr:
                for _ in range(10):
                    ops.feedforward(x, wp, bp, D, w2, None, residual=res, packed64=pk)
                torch.cuda.synchronize()
            for k, (c, ms) in sorted(tr.result.items()):
                if "e2" in k:
                    fl = 2.0 * M * 2 * D * C
                    print(f"  {name:14s} {k:40s} {1e3 * ms / c:8.1f} us  {fl / (ms / c * 1e-3) / 1e12:7.1f} TFLOP/s")
