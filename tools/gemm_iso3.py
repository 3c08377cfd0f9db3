"""Three contraction shapes in isolation across several builds of the library, interleaved in ONE process per build order
(developer tool): python tools/gemm_iso3.py lib1.so lib2.so ...  ('-' = the in-tree library).  Each build runs in its own
subprocess (one library per process); rounds alternate so that clock drift hits every build alike."""
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
    for M, N, K, geglu in [(64512, 5120, 640, 0), (16128, 10240, 1280, 0), (16128, 1280, 5120, 0), (64512, 2560, 640, 1), (64512, 640, 2560, 0)]:
        x = torch.randn(M, K, device=dev).to(H)
        if geglu:
            wp, bp, _ = ops.pack_geglu((torch.randn(2 * N, K, device=dev) * K ** -0.5).to(H), torch.randn(2 * N, device=dev).to(H))
            f = lambda: ops.linear_geglu(x, wp, bp, N)
            fl = 2.0 * M * 2 * N * K
        else:
            w = (torch.randn(N, K, device=dev) * K ** -0.5).to(H)
            f = lambda: ops.linear(x, w)
            fl = 2.0 * M * N * K
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 30)
        t = sorted(ts)[2]
        out.append(f"{'geglu' if geglu else 'dense'} M{M} N{N} K{K}: {t * 1e3:7.1f} us {fl / t / 1e9:6.0f} TF")
    print(" | ".join(out))
    sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "-":
        env["SYN3R_LIB_OVERRIDE"] = str((ROOT / lib).resolve())
    r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
    print(f"{lib:24s}", r.stdout.strip() or r.stderr[-500:], flush=True)
