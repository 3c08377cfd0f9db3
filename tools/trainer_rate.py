"""HOT LOOP A as the trainer runs it: iterations / s of GSTrainer.training() at 200 000 Gaussians / 1920x1080 (developer tool).
usage: python tools/trainer_rate.py [iterations] [explicit|autograd]"""
import sys
import tempfile
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import measure

its = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = sys.argv[2] if len(sys.argv) > 2 else "explicit"
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as tmp:
    tr = measure.synthetic_scene(dev, 200_000, 1080, 1920, 2, its, tmp)
    if mode == "autograd":
        step = tr.train_step
        tr.train_step = lambda cam=None: step(cam, explicit=False)
    tr.training(0, iterations=50, disable_densification=True)          # warm-up: capacities, workspaces
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.training(0, iterations=its, disable_densification=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{mode}: {its} iterations in {dt:.2f} s = {its / dt:.1f} it/s (incl. the loop's exact warm-up renders and its checkpoint)")
