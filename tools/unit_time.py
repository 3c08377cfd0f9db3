"""Wall time of one SVD (step, pass) unit, optionally under environment overrides (developer tool):
python tools/unit_time.py [frames] [ENV=VAL ...]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench
F = int(sys.argv[1]) if len(sys.argv) > 1 else 14
b = SvdStepBench(F, torch.device("cuda", 0))
for _ in range(2):
    b.step_pass()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 6
e0.record()
for _ in range(n):
    b.step_pass()
e1.record()
torch.cuda.synchronize()
print(f"{' '.join(sys.argv[2:]) or 'default':28s} F={F}: {e0.elapsed_time(e1) / n:8.2f} ms per unit, peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
