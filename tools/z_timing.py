"""Where a k_gemm_z wavefront spends its cycles (developer tool): s_memtime ticks per k-tile of the loop's segments.  Needs a
timing build:  tools/build_variant.sh timing -DSYN3R_TIMING  and  SYN3R_LIB_OVERRIDE=abtmp/libtiming.so python tools/z_timing.py"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.unet import ops
from syn3r_amd import _lib
dev = torch.device("cuda", 0)
H = torch.float16
lib = ctypes.CDLL(str(_lib._LIB_PATH))
if not hasattr(lib, "syn3r_debug_wide_timing"):
    sys.exit("library built without -DSYN3R_TIMING")
NAMES = ["g0-8(dma)", "g9-16", "lgkm0+g17", "vmwait", "barrier", "g18-19+rd", "epilogue"]
for M, N, K in [(16128, 1280, 5120), (64512, 5120, 640), (64512, 640, 2560)]:
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(H)
    for _ in range(50):
        ops.linear(x, w)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    lib.syn3r_debug_wide_timing(out)
    print(f"M{M} N{N} K{K}: cycles per k-tile")
    for wv in range(8):
        v = [out[wv * 8 + i] for i in range(8)]
        nk = max(v[7], 1)
        print(f"  wave {wv}: " + "  ".join(f"{n}={x / nk:7.1f}" for n, x in zip(NAMES, v[:7])) + f"  sum={sum(v[:7]) / nk:8.1f}  k-tiles={nk}")
