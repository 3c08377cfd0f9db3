"""Scene-parallel path with world_size 2 over gloo on the CPU (the N>1 path of bench.py)."""
import pytest
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
    rec = [float(rank), 20.0 + rank if ok else math.nan, 0.7, 0.2, 400.0 + rank, 5.0, 12.0, 0.0, ok]
    allrec = D.gather_records(rec)
    assert allrec.shape == (world, len(D.RECORD_FIELDS))
    assert allrec[:, 0].tolist() == [float(r) for r in range(world)]
    assert torch.isnan(allrec[1, 1]) and allrec[0, 1] == 20.0
    tab = D.gather_record_table([rec, [math.nan] * (len(D.RECORD_FIELDS) - 1) + [0.0]])
    assert tab.shape == (2 * world, len(D.RECORD_FIELDS)) and tab[2 * rank, 0] == float(rank) and tab[2 * rank + 1, -1] == 0.0
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
    rec = D.gather_records([0, 30.0, 0.9, 0.1, 100.0, 5.0, 1.0, 0.0, 1.0], device=__import__("torch").device("cpu"))
    assert rec.shape == (1, len(D.RECORD_FIELDS)) and D.RECORD_FIELDS[-2:] == ("truncated_renders", "ok")
    assert D.assign_scenes(list("abcdefghij"), 1, 8) == ["b", "j"]


def test_launcher_arguments_and_assignment():
    """the launcher keeps scripts/train.py's hot-path flags (:50-66) and deals scenes round-robin"""
    from syn3r_amd import dist as D
    from syn3r_amd import launch
    a = launch.parse(["--scenes", "fern,flower,fortress", "--diffusion_type", "2PassProbUncertain", "--interp_type", "backward_warp",
                      "--densify_type", "interpolate_loop0_gs", "--refine_cycle_num", "2", "--cam_confidence", "0.05",
                      "--pseudo_cam_sampling_rate", "0.02", "--num_views_for_pcd_densification", "1", "--dataset", "dtu"])
    assert a.refine_cycle_num == 2 and a.densify_type == "interpolate_loop0_gs" and a.cam_confidence == 0.05
    assert D.assign_scenes(a.scenes.split(","), 1, 2) == ["flower"]
    import pytest
    with pytest.raises(SystemExit):
        launch.parse(["--scenes", "x", "--interp_type", "sideways"])


def test_launcher_defaults_are_the_reference_scripts(golden_dir):
    """Every flag scripts/train.py declares itself (:28-69) exists in the launcher with the reference's default, choices,
    type and action - from tests/golden/train_flags.json, the table oracle/gen_golden.py train_flags read off the script."""
    import json
    from syn3r_amd import launch
    rows = json.loads((golden_dir / "train_flags.json").read_text())
    assert len(rows) == 29
    a = launch.parse(["--scenes", "x"])
    types = {"float": float, "int": int, "str": str, "bool": bool}
    import argparse
    ap_actions = {}
    orig = argparse.ArgumentParser.add_argument

    def spy(self, *flags, **kw):
        act = orig(self, *flags, **kw)
        ap_actions[flags[0]] = act
        return act

    argparse.ArgumentParser.add_argument = spy
    try:
        launch.parse(["--scenes", "x"])
    finally:
        argparse.ArgumentParser.add_argument = orig
    for r in rows:
        flag = r["flags"][0]
        assert flag in ap_actions, flag
        act = ap_actions[flag]
        want = r.get("default", False if r.get("action") == "store_true" else None)
        assert getattr(a, flag.lstrip("-")) == want, (flag, getattr(a, flag.lstrip("-")), want)
        assert (list(act.choices) if act.choices else None) == r.get("choices"), flag
        if "type" in r:
            assert act.type is types[r["type"]], flag
        if r.get("action") == "store_true":
            assert isinstance(act, argparse._StoreTrueAction), flag
        else:
            assert act.nargs == r.get("nargs"), flag
    # the batch scripts' command lines carry FSGS flags as well: accepted, reported as ignored
    b = launch.parse("--scenes fern --iteration dgs1 --weight_clamp 0.2 --diffusion_type 2PassProbUncertainPost --interp_type backward_warp "
                     "--cam_confidence 0.05 --pseudo_cam_sampling_rate 0.02 --densify_type interpolate_gs_v2 --refine_cycle_num 2 "
                     "--num_views_for_pcd_densification 1 -s /data/fern --eval --n_views 3 --resolution 1 --use_dust3r 0".split())
    assert b.refine_cycle_num == 2 and b.num_views_for_pcd_densification == 1 and b.weight_clamp == 0.2
    assert b.ignored_flags == ["-s", "/data/fern", "--eval", "--n_views", "3", "--resolution", "1", "--use_dust3r", "0"]
    launch.validate(b)
    # ... but only those: a misspelled flag of ours is an error, not a silently dropped setting
    for bad in (["--iteratons", "500"], ["--lambda-dssim", "0.3"], ["--diffusion-type", "2PassProbUncertain"]):
        with pytest.raises(SystemExit):
            launch.parse(["--scenes", "x"] + bad)
    # the reference's own defaults name types its orchestrator refuses (model/diffusionGS.py:115-124, :244-255): rejected before any
    # GPU work instead of one NaN record per scene
    with pytest.raises(SystemExit, match="diffusion_type"):
        launch.validate(launch.parse(["--scenes", "x"]))
    with pytest.raises(SystemExit, match="densify_type"):
        launch.validate(launch.parse(["--scenes", "x", "--diffusion_type", "2PassProbUncertainPost"]))
    launch.validate(launch.parse(["--scenes", "x", "--diffusion_type", "2PassProbUncertainPost", "--refine_cycle_num", "0"]))
