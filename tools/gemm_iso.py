"""Time single contraction shapes in isolation, cold (20 launches) and sustained (1 s of launches), optionally from
another build of the library (developer tool):  python tools/gemm_iso.py [lib.so|-] [ENV=VAL ...]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
from syn3r_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib._LIB_PATH = Path(sys.argv[1]).resolve()
import torch
from syn3r_amd.unet import ops
dev = torch.device("cuda", 0)
H = torch.float16
SHAPES = [(16128, 1280, 5120), (64512, 640, 2560), (64512, 5120, 640), (258048, 2560, 320), (258048, 960, 320), (16128, 10240, 1280)]


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * 0.02).to(H)
    f = lambda: ops.linear(x, w)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    cold = timed(f, 20)
    n = max(20, int(1000 / cold))
    timed(f, n)
    hot = timed(f, n // 2)
    fl = 2 * M * N * K / 1e9
    print(f"M{M} N{N} K{K}: cold {cold * 1e3:7.1f} us {fl / cold:6.0f} TF | sustained {hot * 1e3:7.1f} us {fl / hot:6.0f} TF")
    del x, w

# implicit-GEMM 3x3 convolutions of the UNet (NHWC): (frames, H, W, Cin, Cout)
for NB, Hh, Ww, Cin, Cout in [(28, 72, 128, 320, 320), (28, 36, 64, 640, 640), (28, 18, 32, 1280, 1280), (28, 72, 128, 640, 320)]:
    x = torch.randn(NB, Hh, Ww, Cin, device=dev).to(H)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.01).to(H)
    f = lambda: ops.conv3x3(x, w)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    cold = timed(f, 20)
    n = max(20, int(500 / cold))
    timed(f, n)
    hot = timed(f, n // 2)
    fl = 2 * NB * Hh * Ww * Cout * 9 * Cin / 1e9
    print(f"conv {NB}x{Hh}x{Ww} {Cin}->{Cout}: cold {cold * 1e3:7.1f} us {fl / cold:6.0f} TF | sustained {hot * 1e3:7.1f} us {fl / hot:6.0f} TF")
    del x, w

# GEGLU projections (the feed-forward's first half): (M, hidden width D, K)
for M, D, K in [(258048, 1280, 320), (64512, 2560, 640), (16128, 5120, 1280)]:
    x = torch.randn(M, K, device=dev).to(H)
    wp, bp, _ = ops.pack_geglu((torch.randn(2 * D, K, device=dev) * K ** -0.5).to(H), torch.randn(2 * D, device=dev).to(H))
    f = lambda: ops.linear_geglu(x, wp, bp, D)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    cold = timed(f, 20)
    n = max(20, int(500 / cold))
    timed(f, n)
    hot = timed(f, n // 2)
    fl = 2 * M * 2 * D * K / 1e9
    print(f"geglu M{M} D{D} K{K}: cold {cold * 1e3:7.1f} us {fl / cold:6.0f} TF | sustained {hot * 1e3:7.1f} us {fl / hot:6.0f} TF")
    del x, wp, bp
