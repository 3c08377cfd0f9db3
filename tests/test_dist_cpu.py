"""Scene-parallel path with world_size 2 over gloo on the CPU (the N>1 path of bench.py)."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

WORKER = textwrap.dedent("""
    import sys, math
    sys.path.insert(0, %r)
    import torch
    from syn3r_amd import dist as D
    rank, world, local = D.init("gloo")
    scenes = ["fern", "flower", "fortress", "horns", "leaves", "orchids", "room", "trex"]
    mine = D.assign_scenes(scenes, rank, world)
    assert mine == scenes[rank::world]
    ok = 1.0 if rank == 0 else 0.0
    rec = [float(rank), 20.0 + rank if ok else math.nan, 0.7, 0.2, 400.0 + rank, 5.0, 12.0, ok]
    allrec = D.gather_records(rec)
    assert allrec.shape == (world, len(D.RECORD_FIELDS))
    assert allrec[:, 0].tolist() == [float(r) for r in range(world)]
    assert torch.isnan(allrec[1, 1]) and allrec[0, 1] == 20.0
    if rank == 0:
        table = D.summary_table(allrec)
        assert "mean over finished scenes" in table
        print("TABLE_OK")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""") % str(ROOT)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world2_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "TABLE_OK" in outs[0]


def test_single_process_gather():
    from syn3r_amd import dist as D
    rec = D.gather_records([0, 30.0, 0.9, 0.1, 100.0, 5.0, 1.0, 1.0], device=__import__("torch").device("cpu"))
    assert rec.shape == (1, 8)
    assert D.assign_scenes(list("abcdefghij"), 1, 8) == ["b", "j"]
