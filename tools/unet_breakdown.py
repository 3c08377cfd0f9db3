"""Per-kernel device time of one SVD (step, pass) unit (developer tool)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
import os
from syn3r_amd import tuning
tuning.from_env()                             # SYN3R_GN_EPILOGUE=0 etc.: host-graph switches for A/B runs (explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench
if os.environ.get("SYN3R_SET_TILE"):          # force a contraction kernel family (syn3r_gemm_set_tile), tuning runs
    L.load().syn3r_gemm_set_tile(int(os.environ["SYN3R_SET_TILE"]))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 14
b = SvdStepBench(F, torch.device("cuda", 0))
b.step_pass(); torch.cuda.synchronize()
n = 3
with L.kernel_trace(detail=len(sys.argv) > 2) as tr:
    for _ in range(n):
        b.step_pass()
    torch.cuda.synchronize()
tot = sum(v[1] for v in tr.result.values())
for k, (c, ms) in sorted(tr.result.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} {c // n:5d} launches  {ms / n:8.3f} ms  {100 * ms / tot:5.1f} %")
print(f"total traced {tot / n:.2f} ms per unit")
