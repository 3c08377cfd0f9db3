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
    assert "mfma_busy" in rf and (rf["mfma_busy"] is None or 0.0 < rf["mfma_busy"] < 1.0)
    cb = b["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    gs = cb["geometry_and_scheduler"]                                # oracle ports beside the device times
    for k in ("W1_forward_warp_576x1024", "W2_inverse_warp_576x1024", "C1_consistency_check_576x1024",
              "S2_step_interp_grad_25x4x72x128", "S3_step_interp_prob_uncertain_25x4x72x128"):
        assert gs[k]["cpu_port_s"] > 0 and gs[k]["gpu_device_us"] > 0, k
    assert b["record_fields"][-2:] == ["truncated_renders", "ok"] and b["per_rank"][0][-2:] == [0.0, 1.0]


def test_two_rank_job_on_one_device(gpu, tmp_path):
    """The N > 1 launch path of bench.py (one process per rank, barrier, one all-gather of the records, max-over-ranks time):
    two ranks sharing this box's single GPU over gloo (the driver runs the real thing on 8 GPUs over RCCL)."""
    import os
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, SYN3R_BENCH_SINGLE_DEVICE="1", SYN3R_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--raster-iters", "5", "--svd", "off", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and len(b["per_rank"]) == 2 and [row[0] for row in b["per_rank"]] == [0.0, 1.0]
    assert all(row[-1] == 1.0 and row[-2] == 0.0 for row in b["per_rank"])
    assert abs(b["value"] - 2 * 5 * 1000.0 / b["ms_per_step"]) < 0.02 * b["value"]       # whole-job iterations / max-over-ranks time


def test_gpus_flag_starts_the_ranks_itself(gpu):
    """`python bench.py --gpus 2` with no torchrun around it (how the driver runs `--gpus 1`): bench.py starts the two ranks
    as a child torchrun job and relays ONE line with n_gpus = 2 (replaces bash_scripts/batch_llff_train.sh:24-47)."""
    import os
    env = dict(os.environ, SYN3R_BENCH_SINGLE_DEVICE="1", SYN3R_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--raster-iters", "5",
                        "--svd", "off", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and len(b["per_rank"]) == 2 and [row[0] for row in b["per_rank"]] == [0.0, 1.0]
    assert abs(b["value"] - 2 * 5 * 1000.0 / b["ms_per_step"]) < 0.02 * b["value"]
