"""Vendor-library yardstick (developer tool, NEVER part of the product path): what hipBLASLt / MIOpen / torch SDPA reach on the
largest contraction shapes of the UNet unit on THIS box, beside this library's kernels on the same tensors, interleaved in one
process (cdna_hip_programming.md rule 24).  The point is the achievable bar at the clock the chip really holds on random data.

    python tools/vendor_yardstick.py > profiles/r04/vendor_yardstick.txt
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as Fn
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
H = torch.float16
PEAK = 2500.0


def timed(f, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ab(fa, fb, flops, label):
    """interleaved rounds: median of 5 rounds of ~50 ms each per arm"""
    for f in (fa, fb):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    n = max(5, int(50.0 / max(timed(fa, 5), 1e-3)))
    ta, tb = [], []
    for _ in range(5):
        ta.append(timed(fa, n))
        tb.append(timed(fb, n))
    ma, mb = sorted(ta)[2], sorted(tb)[2]
    print(f"{label:44s} vendor {ma * 1e3:8.1f} us {flops / ma / 1e9:6.0f} TF ({flops / ma / 1e9 / PEAK:.3f}) | "
          f"syn3r {mb * 1e3:8.1f} us {flops / mb / 1e9:6.0f} TF ({flops / mb / 1e9 / PEAK:.3f}) | syn3r/vendor time {mb / ma:.2f}", flush=True)


print(torch.cuda.get_device_name(0), torch.__version__)
# ---- dense projections (M, N, K): out = x . w^T
for M, N, K in [(64512, 5120, 640), (16128, 10240, 1280), (64512, 640, 2560), (16128, 1280, 5120), (258048, 960, 320),
                (258048, 2560, 320), (258048, 320, 1280), (258048, 320, 320), (64512, 1920, 640), (16128, 3840, 1280),
                (4032, 10240, 1280), (4032, 1280, 5120)]:
    x = torch.randn(M, K, device=dev).to(H)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(H)
    ab(lambda: torch.matmul(x, w.t()), lambda: ops.linear(x, w), 2.0 * M * N * K, f"dense M{M} N{N} K{K}")
    del x, w

# ---- 3x3 convolutions (frames, H, W, Cin, Cout): MIOpen on channels-last fp16 against the implicit-GEMM kernel on NHWC
for NB, Hh, Ww, Cin, Cout in [(28, 72, 128, 320, 320), (28, 36, 64, 640, 640), (28, 18, 32, 1280, 1280), (28, 72, 128, 640, 320),
                              (28, 36, 64, 1280, 640), (28, 18, 32, 2560, 1280), (28, 9, 16, 1280, 1280)]:
    xn = torch.randn(NB, Hh, Ww, Cin, device=dev).to(H)
    wn = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(H)
    xc = xn.permute(0, 3, 1, 2)                                  # NCHW view of NHWC memory = channels_last
    wc = wn.permute(0, 3, 1, 2)
    ab(lambda: Fn.conv2d(xc, wc, padding=1), lambda: ops.conv3x3(xn, wn), 2.0 * NB * Hh * Ww * Cout * 9 * Cin,
       f"conv3x3 {NB}x{Hh}x{Ww} {Cin}->{Cout}")
    del xn, wn, xc, wc

# ---- spatial self-attention (sequences, heads, S, 64): torch SDPA against k_attn_spatial on the packed qkv rows
for nseq, heads, S in [(28, 5, 9216), (28, 10, 2304), (28, 20, 576)]:
    C = heads * 64
    qkv = torch.randn(nseq * S, 3 * C, device=dev).to(H)
    q, k, v = (t.reshape(nseq, S, heads, 64).transpose(1, 2).contiguous() for t in qkv.split(C, dim=1))
    ab(lambda: Fn.scaled_dot_product_attention(q, k, v), lambda: ops.attention(qkv, nseq, S, heads), 4.0 * nseq * heads * S * S * 64,
       f"attention {nseq}x{heads} heads S{S}")
    del qkv, q, k, v
