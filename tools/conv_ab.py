"""3x3 / temporal convolutions of the UNet in isolation on the 160-column tile (SYN3R_CONV_Z=0) against the 256 x 320 tile
(SYN3R_CONV_Z=1), one subprocess per setting (developer tool).
NOTE (round 5): the library's dispatch switches (SYN3R_GEMM_Z, SYN3R_GEMM_WIDE, SYN3R_CONV_Z, SYN3R_Z_BAND, SYN3R_TCONV_ORDER, ...) are
compiled in only with -DSYN3R_TUNING: build the variant first (`tools/build_variant.sh tune -DSYN3R_TUNING`) and point
SYN3R_LIB_OVERRIDE at it; the shipped library ignores the environment."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import torch
    import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
    from syn3r_amd.unet import ops
    dev = torch.device("cuda", 0)
    H = torch.float16
    out = []

    def timed(f):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        return sorted(ts)[2]
    for NB, Hh, Ww, Cin, Cout in [(28, 72, 128, 320, 320), (28, 36, 64, 640, 640), (28, 18, 32, 1280, 1280), (28, 72, 128, 640, 320), (28, 36, 64, 1280, 640), (28, 18, 32, 2560, 1280)]:
        x = torch.randn(NB, Hh, Ww, Cin, device=dev).to(H)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * (9 * Cin) ** -0.5).to(H)
        b = torch.randn(Cout, device=dev).to(H)
        t = timed(lambda: ops.conv3x3(x, w, b))
        out.append(f"conv {Hh}x{Ww} {Cin}->{Cout}: {t * 1e3:7.1f} us {2.0 * NB * Hh * Ww * Cout * 9 * Cin / t / 1e9:5.0f} TF")
        del x, w
    for B, F, HW, C in [(2, 14, 9216, 320), (2, 14, 2304, 640), (2, 14, 576, 1280)]:
        x = torch.randn(B * F * HW, C, device=dev).to(H)
        w = (torch.randn(C, 3, C, device=dev) * (3 * C) ** -0.5).to(H)
        b = torch.randn(C, device=dev).to(H)
        t = timed(lambda: ops.tconv3(x, w, b, B, F, HW))
        out.append(f"tconv HW{HW} C{C}: {t * 1e3:7.1f} us {2.0 * B * F * HW * C * 3 * C / t / 1e9:5.0f} TF")
        del x, w
    print(" | ".join(out))
    sys.exit(0)
for v in ("0", "1", "0", "1"):
    env = dict(os.environ, SYN3R_CONV_Z=v)
    r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
    print(f"SYN3R_CONV_Z={v}", r.stdout.strip() or r.stderr[-800:], flush=True)
