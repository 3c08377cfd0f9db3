"""Scene-parallel launcher: `DiffusionGS(...).run(refine_cycle_num)` once per scene, one process per GPU, ONE RCCL
all-gather of the per-scene record at the end.

Replaces the reference's sequential bash loop over scenes (`bash_scripts/batch_llff_train.sh:24-47`, one
`python scripts/train.py ...` after the other on a single GPU) and keeps `scripts/train.py`'s hot-path flags
(`:28-69`: --diffusion_type, --interp_type, --cam_confidence, --pseudo_cam_sampling_rate, --densify_type,
--refine_cycle_num, --num_views_for_pcd_densification, --fps_keyframe_sampling, --checkpoint_iterations, --dataset).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m syn3r_amd.launch \\
        --scenes fern,flower,fortress,horns,leaves,orchids,room,trex --scene-factory mypkg.llff:open_scene ...

Rank r takes scenes r, r + world, ... (`dist.assign_scenes`); there is no data-path collective.  What a "scene" is
comes from a factory `f(name, args, device) -> dict(trainer=GSTrainer, num_input_views=int, test_cameras=[Camera],
svd_components=dict|None)`: dataset IO (COLMAP / LLFF readers) is outside this repository's scope, so the built-in
factory is the seeded synthetic scene the tests use (`synthetic:<seed>`), with stand-in SVD modules unless a local
checkpoint directory is given (`--svd_dir`, loaded by `StableVideoDiffusionPipeline.from_pretrained`: UNet, VAE, scheduler, CLIP).
A scene that raises is recorded with NaN metrics and ok = 0; the other ranks are unaffected.
"""
from __future__ import annotations

import argparse
import importlib
import math
import os
import sys
import time
import traceback
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import dist as D


def parse(argv: Optional[Sequence[str]] = None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scenes", type=str, required=True, help="comma-separated scene names")
    ap.add_argument("--scene-factory", type=str, default="syn3r_amd.launch:synthetic_scene",
                    help="module:function building one scene (see the module docstring)")
    ap.add_argument("--model_path", type=str, default="output", help="per-scene artefacts go to <model_path>/<scene>")
    ap.add_argument("--svd_dir", type=str, default=None, help="local SVD checkpoint directory in the diffusers layout (unet/, vae/, scheduler/, image_encoder/, feature_extractor/)")
    ap.add_argument("--backend", type=str, default=None, help="torch.distributed backend (default: nccl = RCCL on a GPU box)")
    # scripts/train.py:28-69, flag for flag with the reference's defaults and choices (tests/golden/train_flags.json, captured
    # from the script by oracle/gen_golden.py train_flags; tests/test_dist_cpu.py compares).  As in the reference the default
    # --diffusion_type '2Pass' and --densify_type 'interpolate' are not among the types the orchestrator accepts
    # (model/diffusionGS.py:115-124, :244-255 raise NotImplementedError): the batch scripts always pass both.
    ap.add_argument("--major_radius", type=float, default=50)           # parsed, unused by the live path
    ap.add_argument("--minor_radius", type=float, default=40)
    ap.add_argument("--weight_clamp", type=float, default=0.4)
    ap.add_argument("--image_path", type=str)
    ap.add_argument("--folder_path", type=str)
    ap.add_argument("--iteration", type=str)
    ap.add_argument("--inverse", type=bool, default=False)
    ap.add_argument("--degrees_per_frame", type=float, default=0.4)
    ap.add_argument("--ip", type=str, default="127.0.0.1")
    ap.add_argument("--port", type=int, default=6007)
    ap.add_argument("--debug_from", type=int, default=-100)
    ap.add_argument("--detect_anomaly", action="store_true", default=False)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--start_checkpoint", type=str, default=None)
    ap.add_argument("--diffusion_type", type=str, default="2Pass")
    ap.add_argument("--interp_type", type=str, default="forward_warp", choices=["forward_warp", "backward_warp"])
    ap.add_argument("--near", type=int, default=0)
    ap.add_argument("--cam_confidence", type=float, default=0.1)
    ap.add_argument("--pseudo_cam_sampling_rate", type=float, default=0.04)
    ap.add_argument("--test_iterations", nargs="+", type=int, default=[5_000, 10_000])
    ap.add_argument("--save_iterations", nargs="+", type=int, default=[5_000, 10_000])
    ap.add_argument("--checkpoint_iterations", nargs="+", type=int, default=[8_000, 10_000])
    ap.add_argument("--train_bg", action="store_true")
    ap.add_argument("--densify_type", type=str, default="interpolate",
                    choices=["interpolate", "from_single", "from_single_gs", "interpolate_gs", "interpolate_loop0_gs", "interpolate_gs_v2"])
    ap.add_argument("--dataset", type=str, default="llff", choices=["llff", "dtu", "dl3dv"])
    ap.add_argument("--refine_cycle_num", type=int, default=1)
    ap.add_argument("--reorg_train_views", type=int, default=1)
    ap.add_argument("--num_views_for_pcd_densification", type=int, default=4)
    ap.add_argument("--fps_keyframe_sampling", type=int, default=0, help="If >0, use FPS for keyframe selection during PCD densification")
    # the trainer's own parameters (FSGS' ModelParams / OptimizationParams groups are absent: the ones this trainer reads)
    ap.add_argument("--iterations", type=int, default=10_000)
    ap.add_argument("--lambda_dssim", type=float, default=0.2)
    ap.add_argument("--lpips_weight", type=float, default=0.0,
                    help="weight of the LPIPS term during refines (bash_scripts/batch_dl3dv_train.sh:84-87 passes 1); needs --lpips_weights")
    ap.add_argument("--lpips_weights", type=str, default=None,
                    help="local state_dict file of lpips.LPIPS(net='vgg') (torch.save format); the package's download is not reachable offline")
    ap.add_argument("--num_inference_steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    # FSGS' parameter groups (scripts/train.py:44-46: -s, --eval, --n_views, --resolution, --use_dust3r ... of the batch scripts)
    # belong to the absent submodule: a command line that carries them is accepted, they are kept in `ignored_flags`
    args, rest = ap.parse_known_args(argv)
    # Only the flags of FSGS' ModelParams / OptimizationParams / PipelineParams groups that the reference's batch scripts pass
    # (bash_scripts/*.sh) are tolerated; anything else is a misspelling of one of ours and an error - a dropped
    # `--iteratons 500` must not start a multi-hour job on the defaults.
    unknown = [t for t in rest if t.startswith("-") and not _is_number(t) and t not in FSGS_FLAGS]
    if unknown:
        ap.error(f"unknown argument(s): {' '.join(unknown)} (flags of the FSGS parameter groups that are accepted and ignored: "
                 f"{' '.join(sorted(FSGS_FLAGS))})")
    args.ignored_flags = list(rest)
    return args


# scripts/train.py:44-46 builds FSGS' three parameter groups on the parser; these are the flags of theirs that the reference's
# bash_scripts/batch_{llff,dtu,dl3dv}_train.sh pass on the command line
FSGS_FLAGS = frozenset([
    "-s", "--source_path", "-m", "-r", "--resolution", "--images", "-i", "--eval", "--n_views", "--use_dust3r", "--rand_pcd",
    "--num_train_samples", "--sample_pseudo_interval", "--sample_svd_pseudo_interval", "--start_sample_svd_frame",
    "--start_sample_pseudo", "--end_sample_pseudo", "--svd_depth_warmup", "--svd_lpips_weight", "--use_proximity_densify",
    "--percent_dense", "--densify_grad_threshold", "--densify_from_iter", "--densify_until_iter", "--densification_interval",
    "--opacity_reset_interval", "--position_lr_init", "--position_lr_final", "--position_lr_max_steps", "--feature_lr",
    "--opacity_lr", "--scaling_lr", "--rotation_lr", "--depth_weight", "--depth_pseudo_weight", "--white_background", "--sh_degree",
    "--data_device", "--convert_SHs_python", "--compute_cov3D_python", "--debug", "--checkpoint", "--video", "-a", "-p", "-f", "-d",
])
# the types the orchestrator accepts (model/diffusionGS.py:115-124, :244-255; syn3r_amd/diffusionGS.py raises on the rest)
DIFFUSION_TYPES = ("2PassProbUncertain", "2PassProbUncertainPost")
DENSIFY_TYPES = ("interpolate_loop0_gs", "interpolate_gs_v2")


def _is_number(t: str) -> bool:
    try:
        float(t)
        return True
    except ValueError:
        return False


def validate(args) -> None:
    """What `parse` cannot reject without departing from the reference's parser (whose DEFAULTS, '2Pass' / 'interpolate', are
    types its own orchestrator refuses): checked by `main` before any GPU work, so a bad command line exits non-zero instead
    of producing one NaN record per scene."""
    if args.diffusion_type not in DIFFUSION_TYPES:              # (DiffusionGS.__init__ raises on it whatever the cycle count)
        raise SystemExit(f"--diffusion_type {args.diffusion_type!r} is not accepted by the orchestrator (model/diffusionGS.py:115-124): "
                         f"pass one of {', '.join(DIFFUSION_TYPES)}")
    if args.refine_cycle_num > 0 and args.densify_type not in DENSIFY_TYPES:
        raise SystemExit(f"--densify_type {args.densify_type!r} is not accepted by the orchestrator (model/diffusionGS.py:244-255): "
                         f"pass one of {', '.join(DENSIFY_TYPES)}")


def synthetic_scene(name: str, args, device) -> dict:
    """`synthetic:<seed>[:<N>]` — a seeded Gaussian cloud rendered from three cameras on a baseline (the views to fit),
    a perturbed copy of it as the initial model, and two held-out cameras between the inputs for PSNR / SSIM."""
    from .gs import Camera, GaussianModel, GSTrainer, OptimizationParams
    from .synthetic import synthetic_gaussians
    parts = name.split(":")
    seed = int(parts[1]) if len(parts) > 1 else 0
    N = int(parts[2]) if len(parts) > 2 else 2000
    H, W = 72, 128
    m, s, q, o, sh = synthetic_gaussians(N, seed=seed, log_scale_mean=math.log(0.08))
    logit = torch.log(o.clamp(1e-3, 1 - 1e-3) / (1 - o.clamp(1e-3, 1 - 1e-3)))
    truth = GaussianModel(m, torch.log(s), q, logit, sh, device=device)
    f = W / (2 * math.tan(math.radians(30)))
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32)

    def pose(dx):
        p = np.eye(4, dtype=np.float32)
        p[0, 3] = dx
        return p

    gt = GSTrainer(truth, [Camera.from_w2c(pose(0.0), K, H, W, data_device=device)])
    shot = lambda dx: gt.render_view(Camera.from_w2c(pose(dx), K, H, W, data_device=device))["render"].detach().clamp(0, 1)
    train = [Camera.from_w2c(pose(dx), K, H, W, image=shot(dx), data_device=device) for dx in (-0.15, 0.0, 0.15)]
    test = [Camera.from_w2c(pose(dx), K, H, W, image=shot(dx), data_device=device) for dx in (-0.075, 0.075)]
    g = torch.Generator().manual_seed(seed + 1)
    model = GaussianModel(m + 0.01 * torch.randn(m.shape, generator=g), torch.log(s), q, logit, sh, device=device)
    opt = OptimizationParams(iterations=args.iterations, lambda_dssim=args.lambda_dssim, seed=args.seed)
    out_dir = os.path.join(args.model_path, name.replace(":", "_"))
    trainer = GSTrainer(model, train, opt, model_path=out_dir, checkpoint_iterations=args.checkpoint_iterations)
    return dict(trainer=trainer, num_input_views=len(train), test_cameras=test, svd_components=None)


def _stand_in_svd(device):
    """Deterministic stand-ins with the interfaces of CLIP / VAE / UNet for runs without a local checkpoint: the loop
    logic, the warps, the scheduler kernels and the rasteriser run for real, the networks do not."""
    class Enc:
        def __call__(self, image):
            from types import SimpleNamespace
            return SimpleNamespace(image_embeds=torch.zeros(1, 1024))

    class Vae:
        config = type("C", (), {"scaling_factor": 0.18215})()

        def encode(self, x):
            from types import SimpleNamespace
            z = torch.nn.functional.avg_pool2d(x, 8)
            z = torch.cat([z, z[:, :1]], 1)
            return SimpleNamespace(latent_dist=SimpleNamespace(mode=lambda: z))

        def decode(self, z, num_frames=1):
            from types import SimpleNamespace
            return SimpleNamespace(sample=torch.nn.functional.interpolate(z[:, :3], scale_factor=8, mode="nearest"))

    class UNet:
        def __call__(self, x, t, encoder_hidden_states=None, added_time_ids=None, return_dict=False):
            return (0.1 * x[:, :, :4],)

    return dict(vae=Vae(), image_encoder=Enc(), unet=UNet(), dtype=torch.float32)


def run_scene(name: str, args, device, factory: Callable) -> List[float]:
    """One scene end to end -> the RECORD_FIELDS record."""
    from .diffusionGS import DiffusionGS
    t0 = time.perf_counter()
    sc = factory(name, args, device)
    trainer = sc["trainer"]
    comps = sc.get("svd_components")
    if comps is None and args.refine_cycle_num > 0:
        # --svd_dir: the whole pipeline from a local diffusers-layout checkpoint (unet/, vae/, scheduler/, image_encoder/,
        # feature_extractor/), loaded by DiffusionGS.svd_render through StableVideoDiffusionPipeline.from_pretrained
        # (model/diffusionGS.py:1089); without it the plumbing stand-ins
        comps = args.svd_dir if args.svd_dir else _stand_in_svd(device)
    trainer.opt.lpips_weight = float(getattr(args, "lpips_weight", 0.0))
    if sc.get("lpips") is not None:
        trainer.lpips = sc["lpips"]
    elif getattr(args, "lpips_weights", None):
        from .gs.lpips import LPIPS
        trainer.lpips = LPIPS().load_state_dict(torch.load(args.lpips_weights, map_location="cpu", weights_only=True), device)
    runner = DiffusionGS(trainer, num_input_views=sc["num_input_views"], save_dir=trainer.scene.model_path,
                         diffusion_type=args.diffusion_type, interp_type=args.interp_type, input_args=args,
                         svd_components=comps, num_inference_steps=args.num_inference_steps,
                         diffusion_size=sc.get("diffusion_size", (576, 1024)))
    it0 = trainer.iteration
    np.random.seed(args.seed)               # the orchestrator's pose perturbation draws from np.random (diffusionGS.py:653)
    runner.run(args.refine_cycle_num)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    ev = trainer.evaluate(sc.get("test_cameras"))
    iters = trainer.iteration - it0
    units = 2.0 * args.num_inference_steps * args.refine_cycle_num * max(sc["num_input_views"] - (args.densify_type == "interpolate_loop0_gs"), 0)
    if trainer.truncated_renders:
        print(f"[syn3r] scene {name}: {trainer.truncated_renders} training render(s) ran with a truncated pair list",
              file=sys.stderr, flush=True)
    return [0.0, ev["psnr"], ev["ssim"], ev["lpips"], iters / wall, units / wall, wall,
            float(trainer.truncated_renders), 1.0]


def main(argv: Optional[Sequence[str]] = None) -> int:
    args = parse(argv)
    validate(args)
    rank, world, local = D.init(args.backend)
    if rank == 0 and args.ignored_flags:
        print(f"[syn3r] flags of the absent FSGS parameter groups ignored: {' '.join(args.ignored_flags)}", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("syn3r_amd.launch needs a HIP device (the SYN3R hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    mod, fn = args.scene_factory.split(":")
    factory = getattr(importlib.import_module(mod), fn)
    scenes = [s for s in args.scenes.split(",") if s]
    mine = D.assign_scenes(scenes, rank, world)
    rounds = math.ceil(len(scenes) / world)
    table = []
    for k in range(rounds):                     # a rank with no scene left contributes a NaN row (ok = 0)
        rec = [float("nan")] * len(D.RECORD_FIELDS)
        rec[-1] = 0.0
        if k < len(mine):
            try:
                rec = run_scene(mine[k], args, device, factory)
            except Exception:                   # scenes are independent: record the failure, keep the job alive
                traceback.print_exc()
                rec = [float("nan")] * len(D.RECORD_FIELDS)
                rec[-1] = 0.0
            rec[0] = float(scenes.index(mine[k]))
        table.append(rec)
    allrec = D.gather_record_table(table).cpu()          # the job's ONE collective (RCCL all-gather over xGMI)
    if rank == 0:
        allrec = allrec[~torch.isnan(allrec[:, 0])]
        allrec = allrec[allrec[:, 0].argsort()]
        print(D.summary_table(allrec))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
