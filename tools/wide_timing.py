"""Where a k_gemm_wide wavefront spends its cycles (developer tool).  Needs a timing build of the library:
    SYN3R_EXTRA_HIPCC_FLAGS=-DSYN3R_TIMING python -m syn3r_amd.build && python tools/wide_timing.py
(and a plain `python -m syn3r_amd.build` afterwards: the instrumentation costs about 10 %).  Prints, per wavefront of
one block, the s_memtime cycles per k-tile of: the DMA wait, the barrier, the DMA issue, and the two halves' fragment
reads and MFMA issue."""
import ctypes, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from syn3r_amd.unet import ops
from syn3r_amd import _lib
dev = torch.device("cuda", 0)
H = torch.float16
lib = ctypes.CDLL(str(_lib._LIB_PATH))
if not hasattr(lib, "syn3r_debug_wide_timing"):
    sys.exit("library built without -DSYN3R_TIMING")
_lib.load().syn3r_gemm_set_tile(-320)     # force the wide tile
NAMES = ["vmwait", "barrier", "dma_issue", "rd0", "mfma0", "rd1", "mfma1"]


def run(M, N, K):
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * 0.02).to(H)
    for _ in range(5):
        ops.linear(x, w)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    rc = lib.syn3r_debug_wide_timing(out)
    nkt = K // 64
    print(f"M{M} N{N} K{K} rc={rc}  (cycles per k-tile)")
    for wv in range(8):
        v = [out[wv * 8 + i] / nkt for i in range(7)]
        print(f"  wave {wv}: " + "  ".join(f"{n}={x:7.1f}" for n, x in zip(NAMES, v)) + f"  total={sum(v):8.1f}")


for shape in [(16128, 1280, 5120), (64512, 5120, 640), (258048, 2560, 320)]:
    run(*shape)
