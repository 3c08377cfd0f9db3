"""Wall-clock of the two halves of a bench step, separately and together (developer tool)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench

sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda", 0)
a = bench.RasterLoop(args, dev)
b = SvdStepBench(args.frames, dev, seed=args.seed)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def raster():
    for _ in range(args.raster_iters):
        a.iteration()


print(f"50 raster iterations: {t(raster):.1f} ms")
print(f"1 SVD unit:           {t(b.step_pass):.1f} ms")
print(f"both (one step):      {t(lambda: (raster(), b.step_pass())):.1f} ms")
