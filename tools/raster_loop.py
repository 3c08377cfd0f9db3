"""A few raster fwd+bwd iterations of the bench scene (developer tool: the target of rocprofv3 --pmc runs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = sys.argv[:1]
args = bench.parse()
loop = bench.RasterLoop(args, torch.device("cuda", 0))
for _ in range(n):
    loop.iteration()
torch.cuda.synchronize()
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import raster
raster.flush_pair_checks()
print("ok")
