"""Visit statistics of the blend kernels on the bench scene (developer tool; needs a -DSYN3R_RASTER_STATS build:
SYN3R_EXTRA_HIPCC_FLAGS=-DSYN3R_RASTER_STATS python -m syn3r_amd.build)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L

sys.argv = sys.argv[:1]
loop = bench.RasterLoop(bench.parse(), torch.device("cuda", 0))
loop.iteration(); torch.cuda.synchronize()
lib = L.load()
buf = (C.c_ulonglong * 4)()
for name in ("fwd", "bwd"):
    getattr(lib, f"syn3r_debug_{name}_stats")(buf, 1)
loop.iteration(); torch.cuda.synchronize()
P = loop.pairs()
for name in ("fwd", "bwd"):
    getattr(lib, f"syn3r_debug_{name}_stats")(buf, 1)
    t, v, a, px = [int(x) for x in buf]
    print(f"{name}: pairs {P}  lane tests {t} ({t / P:.2f}/pair)  wave visits {v} ({v / P:.2f}/pair)  active visits {a} "
          f"({a / max(v, 1):.2f} of visits)  active pixels {px} ({px / max(a, 1):.1f} per active visit of 128)")
