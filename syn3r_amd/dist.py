"""Scene-parallel launch helpers: one process per GPU, rank r owns scene r, no
data-path collective; ONE all-gather of a fixed-size metric record per rank at the end.

The reference runs scenes sequentially from bash (`bash_scripts/batch_llff_train.sh:24-47`) and has
no distributed code at all.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist

# `truncated_renders`: training renders whose (Gaussian, tile) list outgrew the async binning capacity (must be 0)
RECORD_FIELDS = ("scene_id", "psnr", "ssim", "lpips", "raster_iters_per_s", "svd_units_per_s", "wall_s",
                 "truncated_renders", "ok")


def init(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world, local


def assign_scenes(scenes: Sequence[str], rank: int, world: int) -> List[str]:
    """Round-robin: rank r gets scenes r, r + world, ... (8 LLFF scenes on 8 GPUs = one each;
    DL3DV's 10 scenes = two rounds)."""
    return [s for i, s in enumerate(scenes) if i % world == rank]


def gather_records(record: Sequence[float], device: torch.device | None = None) -> torch.Tensor:
    """All-gather one float32 record of len(RECORD_FIELDS) per rank -> [world, fields] on every rank.
    A failed scene reports NaN metrics with ok = 0 (no elastic recovery: scenes are independent)."""
    if len(record) != len(RECORD_FIELDS):
        raise ValueError(f"record must have {len(RECORD_FIELDS)} fields: {RECORD_FIELDS}")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (
            dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    rec = torch.tensor(list(record), dtype=torch.float32, device=device)
    if world == 1:
        return rec[None]
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    return torch.stack(out)


def gather_record_table(records: Sequence[Sequence[float]], device: torch.device | None = None) -> torch.Tensor:
    """All-gather a rank's [k, fields] table of records (k equal on every rank: pad with NaN / ok = 0 rows) in ONE
    collective -> [world * k, fields] on every rank, rank-major.  This is the job's only collective."""
    rows = [list(r) for r in records]
    if any(len(r) != len(RECORD_FIELDS) for r in rows):
        raise ValueError(f"every record must have {len(RECORD_FIELDS)} fields: {RECORD_FIELDS}")
    world = dist.get_world_size() if dist.is_initialized() else 1
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if (
            dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    tab = torch.tensor(rows, dtype=torch.float32, device=device).reshape(len(rows), len(RECORD_FIELDS))
    if world == 1:
        return tab
    out = [torch.empty_like(tab) for _ in range(world)]
    dist.all_gather(out, tab)
    return torch.cat(out)


def summary_table(records: torch.Tensor) -> str:
    """`scripts/summarize_dl3dv.py`-style table of the gathered records."""
    rows = ["  ".join(f"{f:>18s}" for f in RECORD_FIELDS)]
    for r in records.tolist():
        rows.append("  ".join(f"{v:18.4f}" for v in r))
    ok = records[records[:, -1] > 0.5]
    if len(ok):
        rows.append("  ".join(f"{v:18.4f}" for v in ok.mean(0).tolist()) + "   (mean over finished scenes)")
    return "\n".join(rows)
