"""Which torch operators (as opposed to the library's own kernels) run inside one SVD unit and one raster iteration
(developer tool): python tools/torch_ops_count.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from torch.profiler import profile, ProfilerActivity
import bench as B
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench

dev = torch.device("cuda", 0)


def report(name, fn, n):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    print(f"== {name} (per call, {n} calls profiled)")
    ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
    mine = ("k_", "(anonymous namespace)", "void (anonymous", "void syn3r", "syn3r::")
    rows = [e for e in ev if not e.key.startswith(mine)][:30]      # torch's own operators and the runtime's copies
    for e in rows:
        print(f"  {e.key[:60]:60s} calls {e.count / n:8.1f}  device {e.device_time_total / n / 1e3:8.3f} ms  cpu {e.cpu_time_total / n / 1e3:8.3f} ms")


b = SvdStepBench(14, dev)
report("SVD unit", b.step_pass, 2)
import argparse
args = argparse.Namespace(gaussians=200_000, height=1080, width=1920, seed=1234, loss="l1")
loop = B.RasterLoop(args, dev)
loop.iteration(); loop.iteration()
report("raster iteration", loop.iteration, 10)
