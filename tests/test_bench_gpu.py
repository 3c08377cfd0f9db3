"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys, a roofline object for the dominant
kernel family measured live, and internally consistent numbers (a short run: 1 step of 5 raster iterations + 1 SVD unit)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_bench_line_contract(gpu):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--raster-iters", "5",
                        "--no-sub-benchmarks"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # ONE line on stdout
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 1 and b["warmup"] == 1 and b["higher_is_better"] is True
    assert b["scaling"] == "weak" and b["vs_baseline"] is None and b["data"] == "synthetic" and "workload" in b["config"]
    assert abs(b["value"] - 5 * 1000.0 / b["ms_per_step"]) < 0.02 * b["value"]          # iterations of the whole job / its time
    rf = b["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    cb = b["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
