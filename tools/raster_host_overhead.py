"""How long does the HOST take to enqueue one raster fwd+bwd iteration? (developer tool)"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import raster
args = types.SimpleNamespace(gaussians=200000, height=1080, width=1920, seed=1234)
dev = torch.device("cuda", 0)
loop = bench.RasterLoop(args, dev)
for _ in range(5):
    loop.iteration()
torch.cuda.synchronize()
import cProfile, pstats
for mode in ("async", "sync"):
    raster.set_pair_count_mode(mode)
    for _ in range(3): loop.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        loop.iteration()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host enqueue {1e3*(t1-t0)/50:.3f} ms/iter, wall {1e3*(t2-t0)/50:.3f} ms/iter")
raster.set_pair_count_mode("async")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): loop.iteration()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
