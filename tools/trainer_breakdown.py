"""Per-kernel device time of one GSTrainer.train_step (explicit step) at 200 000 Gaussians / 1920x1080 (developer tool).
usage: python tools/trainer_breakdown.py [iterations] [densify]"""
import sys
import tempfile
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L, measure, raster

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
densify = len(sys.argv) > 2 and sys.argv[2] == "densify"          # with the per-iteration densification statistics
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as tmp:
    tr = measure.synthetic_scene(dev, 200_000, 1080, 1920, 2, 1000, tmp)
    tr.training(0, iterations=50, disable_densification=True)          # warm-up: capacities, workspaces
    tr.densify = densify
    tr.opt.densify_from_iter = 10 ** 9                               # statistics only: the set of Gaussians stays fixed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        tr.train_step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 100
    with L.kernel_trace() as k:
        for _ in range(n):
            tr.train_step()
        torch.cuda.synchronize()
    raster.flush_pair_checks()
tot = sum(v[1] for v in k.result.values())
for name, (c, ms) in sorted(k.result.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} {c / n:6.1f} launches  {1e3 * ms / n:8.1f} us  {100 * ms / tot:5.1f} %")
print(f"traced kernels {1e3 * tot / n:.1f} us per iteration (torch's own kernels are not traced); wall {1e6 * wall:.1f} us per iteration ({1 / wall:.0f} it/s)")
