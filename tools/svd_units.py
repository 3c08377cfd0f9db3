"""Time one (step, pass) unit of both pipeline variants on synthetic inputs (developer tool, GPU box)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import SvdStepBench
from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline

dev = torch.device("cuda", 0)
for F in (14, 25):
    b = SvdStepBench(F, dev)
    pipe = StableVideoDiffusionPipeline(None, None, b.unet, b.sch, variant="post", device=dev)
    pipe._guidance_scale = b.guidance
    b.sch.set_timesteps(100)
    t = b.sch.timesteps[10]
    for name, fn in (("replace", lambda: b.step_pass()),
                     ("post", lambda: pipe._pass_post(10, t, b.latents, b.image_latents, b.ehs, b.added, b.temp_cond,
                                                      b.mask, b.lambda_ts, True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"F={F} {name}: {1e3 * (time.perf_counter() - t0) / n:.1f} ms per (step, pass) unit", flush=True)
    del b, pipe
    torch.cuda.empty_cache()
