"""Per-kernel / per-shape device time of BOTH passes of one denoising step stacked into one launch sequence (merge_passes):
python tools/unet_breakdown_merged.py [F] [replace|post]   (developer tool)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import _lib as L
from syn3r_amd.pipeline.svd_step import SvdStepBench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 25
variant = sys.argv[2] if len(sys.argv) > 2 else "replace"
b = SvdStepBench(F, torch.device("cuda", 0))
b.step_both(variant); torch.cuda.synchronize()
n = 2
with L.kernel_trace(detail=True) as tr:
    for _ in range(n):
        b.step_both(variant)
    torch.cuda.synchronize()
tot = sum(v[1] for v in tr.result.values())
print(f"{variant}, F = {F}: both passes of a step in one stack of launches = 2 (step, pass) units; per-step figures")
for k, (c, ms) in sorted(tr.result.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} {c // n:5d} launches  {ms / n:8.3f} ms  {100 * ms / tot:5.1f} %")
print(f"total traced {tot / n:.2f} ms per step = {tot / n / 2:.2f} ms per (step, pass) unit")
