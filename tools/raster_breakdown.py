"""Per-kernel device time of the bench's raster iteration (developer tool):
python tools/raster_breakdown.py [iters] [l1|l1+ssim] [height width [gaussians]]   (e.g. 20 l1 378 504: an LLFF / DTU-size image)"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L, raster

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
loss = sys.argv[2] if len(sys.argv) > 2 else "l1"
extra = []
if len(sys.argv) > 4:
    extra = ["--height", sys.argv[3], "--width", sys.argv[4]] + (["--gaussians", sys.argv[5]] if len(sys.argv) > 5 else [])
sys.argv = sys.argv[:1] + ["--loss", loss] + extra
args = bench.parse()
loop = bench.RasterLoop(args, torch.device("cuda", 0))
for _ in range(5):
    loop.iteration()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    loop.iteration()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 50
with L.kernel_trace() as tr:
    for _ in range(n):
        loop.iteration()
    torch.cuda.synchronize()
raster.flush_pair_checks()
tot = sum(v[1] for v in tr.result.values())
for k, (c, ms) in sorted(tr.result.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {c / n:6.1f} launches  {1e3 * ms / n:8.1f} us  {100 * ms / tot:5.1f} %")
print(f"traced kernels {1e3 * tot / n:.1f} us per iteration; untraced wall {1e3 * wall:.1f} us per iteration ({1 / wall:.0f} it/s)")
