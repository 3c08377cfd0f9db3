"""Developer tool: LPIPS forward + backward at the bench resolution (1080p): ms per call and per kernel family."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
from syn3r_amd.gs.lpips import LPIPS

dev = torch.device("cuda", 0)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
m = LPIPS().init_random(dev)
a = torch.rand(3, H, W, device=dev).requires_grad_(True)
b = torch.rand(3, H, W, device=dev)
for _ in range(2):
    m(a, b).backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    m(a, b).backward()
torch.cuda.synchronize()
print(f"LPIPS fwd+bwd at {W}x{H}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per call (target features cached)")
with L.kernel_trace() as tr:
    m(a, b).backward()
    torch.cuda.synchronize()
for k, (c, ms) in sorted(tr.result.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} {c:4d} launches {ms:8.3f} ms")
