"""Developer tool: cProfile of ONE view pair through the orchestrator (everything but svd_render) at the bench scene size."""
import cProfile, pstats, sys, tempfile
from pathlib import Path
from types import SimpleNamespace
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd import measure as M
from syn3r_amd.diffusionGS import DiffusionGS

dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as tmp:
    trainer = M.synthetic_scene(dev, 200_000, 1080, 1920, 2, 10, tmp)
    args = SimpleNamespace(cam_confidence=0.05, pseudo_cam_sampling_rate=0.02, fps_keyframe_sampling=0,
                           densify_type="interpolate_loop0_gs", num_views_for_pcd_densification=1)
    d = DiffusionGS(trainer, num_input_views=2, save_dir=tmp, diffusion_type="2PassProbUncertainPost", interp_type="backward_warp",
                    input_args=args, svd_components=dict(vae=None), num_inference_steps=2)
    d.svd_render = lambda image_l, image_r, masks, cond_image, output_path, lambda_ts, num_frames=25, save_prefix="": \
        [np.zeros((576, 1024, 3), np.float32) for _ in range(num_frames)]
    np.random.seed(0)
    d._interpolate_between_gs_v3(0, 1)          # warm-up
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    d._interpolate_between_gs_v3(0, 1)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
