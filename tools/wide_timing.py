"""Where a k_gemm_wide wavefront spends its cycles (developer tool).  Needs a timing build of the library:
    SYN3R_EXTRA_HIPCC_FLAGS=-DSYN3R_TIMING python -m syn3r_amd.build && python tools/wide_timing.py
(and a plain `python -m syn3r_amd.build` afterwards: the instrumentation costs about 10 %).  Prints, per wavefront of
one block, the s_memtime cycles per k-tile of: the DMA wait, the barrier, the DMA issue, and the two halves' fragment
reads and MFMA issue."""
import ctypes, os, sys
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


def run_persistent(M, N, K, geglu):
    """k_gemm_widep: ticks per tile of (top: setup + first wait), (k-loop), (epilogue), and the in-kernel clock."""
    _lib.load().syn3r_gemm_set_tile(0)
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * 0.02).to(H)
    b = torch.zeros(N, device=dev, dtype=H)
    if geglu:
        wp, bp, _D = ops.pack_geglu(w, b)
        f = lambda: ops.linear_geglu(x, wp, bp, N // 2)
    else:
        f = lambda: ops.linear(x, w)
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    lib.syn3r_debug_wide_timing(out)
    print(f"persistent M{M} N{N} K{K} geglu={geglu}")
    for wv in (0, 4):
        top, loop, epi, nt, mt, rt, k0, k1 = [out[wv * 8 + i] for i in range(8)]
        nt = max(nt, 1)
        nkt = K // 64
        if geglu:      # slot 7 carries the gate's share of the epilogue instead of k-tile 1
            print(f"           k-tile 0: {k0 / nt:7.0f}  gate (barrier + bias + GELU arithmetic): {k1 / nt:7.0f} of the epilogue")
        else:
            print(f"           k-tile 0: {k0 / nt:7.0f}  k-tile 1: {k1 / nt:7.0f}  k-tiles 2..: {(loop - k0 - k1) / nt / max(nkt - 2, 1):7.0f} each")
        print(f"  wave {wv}: tiles {nt}  per tile: top {top / nt:8.0f}  k-loop {loop / nt:8.0f}  epilogue {epi / nt:8.0f} ticks;"
              f"  clock {mt / max(rt, 1) * 100:.0f} MHz ({mt} memtime / {rt} realtime ticks)")


def run_ffn():
    """k_ffn320: ticks per 64-wide hidden chunk of each segment (one block, wavefronts 0 and 4)."""
    M, C, D = 258048, 320, 1280
    x = torch.randn(M, C, device=dev).to(H)
    w1 = (torch.randn(2 * D, C, device=dev) * 0.05).to(H); b1 = torch.zeros(2 * D, device=dev, dtype=H)
    w2 = (torch.randn(C, D, device=dev) * 0.03).to(H); b2 = torch.zeros(C, device=dev, dtype=H)
    cw, cb = ops.pack_geglu_chunked(w1, b1)[:2]
    for _ in range(30):
        ops.feedforward_fused(x, cw, cb, D, w2, b2, residual=x)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    lib.syn3r_debug_wide_timing(out)
    names = ["p1 wait", "p1 barrier", "p1 reads+issue", "p1 mfma", "gate", "P wait+barrier", "phase 2"]
    for wv in (0, 4):
        n = max(out[wv * 8 + 7], 1)
        v = [out[wv * 8 + i] / n for i in range(7)]
        print(f"  ffn320 wave {wv}: " + "  ".join(f"{a}={b:7.0f}" for a, b in zip(names, v)) + f"  per chunk total={sum(v):8.0f} (matrix work 3840)")


if "--ffn" in sys.argv:
    run_ffn()
elif "--persistent" in sys.argv:
    for M, N, K, g in [(64512, 640, 2560, False), (64512, 1920, 640, False), (16128, 3840, 1280, False), (16128, 1280, 5120, False)]:
        run_persistent(M, N, K, g)
else:
    for shape in [(16128, 1280, 5120), (64512, 5120, 640), (258048, 2560, 320)]:
        run(*shape)
