"""Developer tool: the F-frame one-pass unit eager against a replayed hipGraph of the UNet's launch sequence."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import tuning; tuning.from_env()      # developer tool: host-graph switches from SYN3R_* variables
from syn3r_amd.pipeline.svd_step import SvdStepBench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = torch.device("cuda", 0)
b = SvdStepBench(F, dev)


def wall_ms(fn, n=10):
    fn(); fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


b.i = 0
y0 = b.step_pass()
print(f"F={F} eager unit   {wall_ms(b.step_pass):.2f} ms")
b.use_graphs = True
b.i = 0
y1 = b.step_pass()
print(f"F={F} graphed unit {wall_ms(b.step_pass):.2f} ms   bit-identical first step: {bool(torch.equal(y0, y1))}")
