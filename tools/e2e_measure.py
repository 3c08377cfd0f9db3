"""Developer tool: the measured end-to-end runs of syn3r_amd/measure.py (full-size svd_render per variant, host gap of the
denoising loop, the scaled schedule through DiffusionGS.run)."""
import json, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import measure as M

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
what = sys.argv[2].split(",") if len(sys.argv) > 2 else ["gap", "replace", "post", "schedule"]
comps = M.full_size_components(dev)
if "gap" in what:
    for v in ("replace", "post"):
        print(json.dumps(M.measure_denoise_gap(comps, v, dev)), flush=True)
for v in ("replace", "post"):
    if v in what:
        print(json.dumps(M.measure_svd_render(comps, v, dev, steps=steps)), flush=True)
if "schedule" in what:
    with tempfile.TemporaryDirectory() as tmp:
        print(json.dumps(M.measure_schedule(comps, dev, tmp, steps=steps)), flush=True)
