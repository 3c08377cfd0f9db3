#!/usr/bin/env python3
"""bench.py — SYN3R hot loop on MI355X: LLFF 3-view train-loop iterations per second.

One bench STEP is one block of the fern-like schedule:
  HOT LOOP A  `--raster-iters` Gaussian-raster forward+backward iterations (L1 loss) at
              N = 200 000 Gaussians, 1920x1080, SH degree 3, and
  HOT LOOP B  one SVD (denoise-step, pass) unit: CFG UNet forward on [2,F,8,72,128] fp16 latents
              + the fused modified-Euler step (F = 14, BASELINE.json configs[1]).
The schedule 10k + 2 x (3 svd_render + 10k) has 30 000 raster iterations per 600 one-pass
(step, pass) units = 50 : 1, which is the default ratio.  value = raster iterations / s over the
whole job (all ranks), the SVD units being amortised inside the same wall-clock.

Launch: `python bench.py` (1 GPU), `python bench.py --gpus N` (starts the N ranks itself: a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, rank 0's JSON line relayed), or that torchrun
command directly (one scene per rank, weak scaling, a single RCCL all-gather of the per-rank metric record
at the end).  The N-rank job replaces the reference's sequential scene loop, bash_scripts/batch_llff_train.sh:24-47.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
# vector-instruction issue: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 4 cycles (MI355X_MICROARCH.md, cycle
# constants: v_fma_f32 "one wave alone: 4") at the 2.4 GHz maximum clock = 614.4 G wave-instructions / s
VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 4.0   # = 614.4 G wave-instructions / s
# ... and what the chip MEASURES for that roof: tools/ubench/valu_rate.hip, raw output in profiles/r06/valu_rate.txt - 8 independent
# v_fma_f32 per iteration, 1 / 2 / 4 wavefronts per SIMD: 3.88 / 2.02 / 2.11 ns per wave-instruction and SIMD (2.06 beside an MFMA).  One
# wavefront alone is latency-bound; from two wavefronts on a wave64 VALU instruction occupies its SIMD for 2.02-2.11 ns = 4 cycles at the
# 1.9-2.0 GHz the chip holds under that load - 4 cycles, not the 2 of the guide's SIMD-32 row.  1024 SIMDs / 2.02 ns:
VALU_ISSUE_MEASURED_GINST = 1024 / 2.02       # = 506.9 G wave-instructions / s (profiles/r06/valu_rate.txt, best of the 2 / 4 wavefront rows)
# SQ_INSTS_VALU per launch of the blend kernels on the bench scene (200 000 Gaussians, 1920x1080, seed 0), used when the committed
# PMC summary of THIS build carries none: profiles/r05/pmc/pmc_counters_by_kernel.json (the same counts as round 4's: the blend
# loops are unchanged, round 5 changed the order in which the tiles are taken and how their lists are built)
VALU_INSTS_FALLBACK = {"k_render_bwd": (200413989.0, "profiles/r05/pmc/pmc_counters_by_kernel.json (an earlier build of this round; blend loops unchanged since)"),
                       "k_render": (69256554.0, "profiles/r05/pmc/pmc_counters_by_kernel.json (an earlier build of this round; blend loops unchanged since)")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--raster-iters", type=int, default=50, help="raster fwd+bwd iterations per step")
    ap.add_argument("--frames", type=int, default=14, help="SVD frames (14 = BASELINE configs[1], 25 = reference)")
    ap.add_argument("--svd", choices=["on", "off"], default="on")
    ap.add_argument("--loss", choices=["l1", "l1+ssim"], default="l1",
                    help="photometric loss of the raster iteration (SURVEY 8d defines the composite with L1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-benchmarks", action="store_true",
                    help="skip the extra timed sub-results (Post / Replace units at F = 25, full trainer iteration, inverse warp)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the measured full-size svd_render calls and the scaled DiffusionGS.run schedule (~2 min of GPU)")
    ap.add_argument("--e2e-steps", type=int, default=100, help="denoising steps of the measured svd_render calls (reference: 100)")
    ap.add_argument("--e2e-iterations", type=int, default=2500,
                    help="trainer iterations of each of the two loops of the scaled schedule (2 x 2500: fixed costs - exact warm-up renders, "
                         "checkpoints - amortised as in a 10 000-iteration loop)")
    ap.add_argument("--no-kernel-trace", action="store_true",
                    help="skip the HIP-event kernel timing (rocprofv3 --pmc passes: the counters serialise every launch)")
    ap.add_argument("--trace-steps", type=int, default=1,
                    help="how many of the timed steps (the last ones) carry the HIP-event kernel timing")
    ap.add_argument("--seed", type=int, default=1234)
    return ap.parse_args()


class RasterLoop:
    """HOT LOOP A: render, L1 loss against a fixed target, backward (no optimiser state changes so
    that every iteration does identical work)."""

    def __init__(self, args, dev):
        from syn3r_amd import synthetic as RO
        from syn3r_amd import raster
        from syn3r_amd.raster import GaussianRasterizationSettings, GaussianRasterizer
        raster.set_pair_count_mode("async")      # training-loop mode: no host round trip per render (checked below)
        m, s, q, o, sh = RO.synthetic_gaussians(args.gaussians, seed=args.seed)
        view, proj, campos, tfx, tfy = RO.look_at_camera(args.height, args.width)
        f = lambda t: t.to(dev).requires_grad_(True)
        self.p = dict(m=f(m), s=f(s), q=f(q), o=f(o), sh=f(sh))
        self.m2 = torch.zeros(args.gaussians, 3, device=dev, requires_grad=True)
        st = GaussianRasterizationSettings(args.height, args.width, tfx, tfy, torch.zeros(3, device=dev), 1.0,
                                           view.to(dev), proj.to(dev), 3, campos.to(dev), False, False)
        self.rast = GaussianRasterizer(st)
        g = torch.Generator().manual_seed(args.seed)
        self.target = torch.rand(3, args.height, args.width, generator=g).to(dev)
        self.N, self.H, self.W = args.gaussians, args.height, args.width
        self.loss_kind = args.loss
        self.P = 0

    def iteration(self):
        from syn3r_amd.gs.train_ops import l1_loss, photometric_loss
        p = self.p
        color, radii, depth, alpha = self.rast(p["m"], self.m2, p["o"], shs=p["sh"], scales=p["s"], rotations=p["q"])
        loss = l1_loss(color, self.target) if self.loss_kind == "l1" else photometric_loss(color, self.target, 0.2)
        loss.backward()
        for t in list(p.values()) + [self.m2]:
            t.grad = None
        return loss

    def full_iteration_rate(self, iters: int = 50) -> float:
        """A COMPLETE trainer iteration as `GSTrainer.train_step` runs it (render -> 0.8 L1 + 0.2 (1 - SSIM), FSGS's default
        lambda_dssim -> backward -> fused Adam step on all five parameter groups), no host synchronisation inside:
        iterations per second.  Runs on copies of the parameters (the headline loop's inputs stay fixed)."""
        from syn3r_amd.gs.train_ops import FusedAdam, photometric_loss
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in self.p.items()}
        # FSGS's learning rates x 1e-3: the synthetic target is noise, at full rates the Gaussians swell and the pair count
        # (the work per iteration) triples within 50 steps - the timing should be of a steady scene
        opt = FusedAdam([{"params": [ps["m"]], "lr": 1.6e-7}, {"params": [ps["sh"]], "lr": 2.5e-6}, {"params": [ps["o"]], "lr": 5e-5},
                         {"params": [ps["s"]], "lr": 5e-6}, {"params": [ps["q"]], "lr": 1e-6}], eps=1e-15)

        def one():
            color, _, _, _ = self.rast(ps["m"], self.m2, ps["o"], shs=ps["sh"], scales=ps["s"], rotations=ps["q"])
            loss = photometric_loss(color, self.target, 0.2)
            opt.zero_grad(set_to_none=True)
            self.m2.grad = None
            loss.backward()
            opt.step()

        for _ in range(3):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
        return iters / (time.perf_counter() - t0)

    def metrics(self):
        """device-side PSNR / SSIM of the current render against the loop's target (the psnr / ssim columns of the
        per-scene record; the synthetic target is noise, so the values only exercise the path)"""
        from syn3r_amd.gs.train_ops import image_metrics
        p = self.p
        with torch.no_grad():
            color, _, _, _ = self.rast(p["m"].detach(), self.m2.detach(), p["o"].detach(), shs=p["sh"].detach(),
                                       scales=p["s"].detach(), rotations=p["q"].detach())
            m = image_metrics(color.clamp(0, 1), self.target)
        return float(m[0]), float(m[1])

    def pairs(self):
        """number of (Gaussian, tile) pairs of this scene/camera (for the roofline byte count)"""
        import ctypes as C
        from syn3r_amd import _lib as L
        lib = L.load()
        p = self.p
        dev = p["m"].device
        geom = torch.empty(lib.syn3r_raster_geom_bytes(self.N), dtype=torch.uint8, device=dev)
        radii = torch.empty(self.N, dtype=torch.int32, device=dev)
        s = self.rast.raster_settings
        P = C.c_longlong(0)
        h16 = lambda m: L.host_f32(m.detach().cpu().reshape(-1).tolist())
        rc = lib.syn3r_raster_preprocess(self.N, 3, 16, L.ptr(p["m"].detach()), L.ptr(p["s"].detach()),
                                         L.ptr(p["q"].detach()), L.ptr(p["o"].detach()), L.ptr(p["sh"].detach()), None,
                                         1.0, h16(s.viewmatrix), h16(s.projmatrix), h16(s.campos), s.tanfovx, s.tanfovy,
                                         self.H, self.W, L.ptr(radii), L.ptr(geom), geom.numel(), C.byref(P),
                                         L.stream_ptr(dev))
        L.check(rc, "preprocess")
        return int(P.value)


def raster_algorithmic_bytes(N, P, H, W):
    """SURVEY.md §8d raster roofline, per kernel (bytes one launch must move at minimum)."""
    hw = H * W
    return {
        "k_preprocess": N * 236 + N * (48 + 4 + 8 + 24 + 16 + 12 + 4 + 4),
        "k_render": P * (4 + 36) + hw * 20 + hw * 8,
        "k_render_bwd": P * (4 + 36) + hw * (20 + 8) + P * 40,
        "k_preprocess_bwd": N * 236 * 2 + N * 64,
        "k_scatter": P * 12 * 2,
        "k_hist": P * 8,
        "k_dup_keys": N * 20 + P * 12,
    }


def cpu_baseline_raster(args):
    """Oracle (CPU restatement, kind='port') on a bounded sample: same 200k-Gaussian scene rendered
    fwd+bwd at 1/2 resolution per axis (about 15 s of CPU work); pixel work is scaled by 4 to the full frame."""
    from oracle import raster_oracle as RO
    # the GPU box exposes many host cores but grants ~16 to a 1-GPU job: oversubscribing them stalls for minutes
    ncores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(ncores)
    N = args.gaussians
    H, W = args.height // 2, args.width // 2
    m, s, q, o, sh = RO.synthetic_gaussians(N, seed=args.seed)
    # keep the footprint in pixels comparable: scales shrink with the image
    view, proj, campos, tfx, tfy = RO.look_at_camera(H, W)
    ps = [t.clone().requires_grad_(True) for t in (m, s, q, o, sh)]
    t0 = time.time()
    color, radii, depth, alpha, aux = RO.rasterize(ps[0], ps[1], ps[2], ps[3], ps[4], None, view, proj, campos, tfx, tfy,
                                                   H, W, torch.zeros(3), 3)
    color.abs().mean().backward()
    dt = time.time() - t0
    full = dt * 4.0
    return dict(value=1.0 / full, unit="iters/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle/raster_oracle.py fwd+bwd, {N} Gaussians at {W}x{H} (1/4 of the pixels) took "
                       f"{dt:.1f} s; scaled x4 to {args.width}x{args.height}; raster iterations only (no CPU UNet: see DESIGN.md)")


def cpu_baseline_unet(args):
    """Oracle (oracle/unet_oracle.py, torch fp32 on the host cores, kind='port') on a bounded sample of HOT LOOP B:
    the FULL SVD-XT UNet configuration (1.52 B seeded weights) on a CFG batch of 2 frames at 72x64 latents (1/14 of the
    unit's tokens, the full latent height); the time is scaled to the benchmark unit by the token ratio (every contraction
    is linear in B*F*h*w; the spatial attention's quadratic term is under-counted at half the width, so this still
    over-states the CPU's throughput)."""
    from oracle.unet_oracle import UNetOracle
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel      # parameter table only
    shapes = UNetSpatioTemporalConditionModel().parameter_shapes()
    g = torch.Generator().manual_seed(args.seed)
    pool = torch.randn(1 << 22, generator=g)      # timing only: the 1.5 B weights are tiled from 4 M normals
    sd = {}
    for k, shape in shapes.items():
        if k.endswith("mix_factor"):
            sd[k] = torch.full(shape, 0.5)
        elif (".norm" in k and k.endswith(".weight")) or k == "conv_norm_out.weight":
            sd[k] = torch.ones(shape)
        elif k.endswith(".bias"):
            sd[k] = torch.zeros(shape)
        else:
            n = math.prod(shape)
            sd[k] = (pool.repeat((n + pool.numel() - 1) // pool.numel())[:n].view(shape)
                     * min(0.02, math.prod(shape[1:]) ** -0.5))
    orc = UNetOracle(sd, {})
    B, F, h, w = 2, 2, 72, 64                      # 18 432 tokens = 1/14 of the benchmark unit's [2,14,8,72,128]
    x = torch.randn(B, F, 8, h, w, generator=g)
    ehs = torch.randn(B, 1, 1024, generator=g)
    added = torch.tensor([[6.0, 127.0, 0.02]] * B)
    t0 = time.time()
    y = orc.forward(x, 1.6378, ehs, added)
    dt = time.time() - t0
    assert bool(torch.isfinite(y).all())
    scale = (2 * args.frames * 72 * 128) / float(B * F * h * w)
    return dt * scale, f"oracle/unet_oracle.py CFG forward [2,{F},8,{h},{w}] (full 1.52 B-parameter configuration) took {dt:.1f} s; x{scale:.0f} by tokens to [2,{args.frames},8,72,128]"


def cpu_baseline_geometry_scheduler(dev):
    """the geometry functions at 576x1024 (SURVEY 8d synthetic warp) and the scheduler steps on
    [1,25,4,72,128], the oracle (kind 'port': numpy restatements pinned by reference-run fixtures) timed on the host cores
    once each, beside the device time of the HIP path on the same inputs (sum of kernel time, HIP events)."""
    from oracle import golden_inputs as GI
    from oracle import scheduler_oracle as SO
    from oracle import warp_oracle as WO
    from syn3r_amd import _lib as L
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.solver_utils.consistency import consistency_check_with_depth
    from syn3r_amd.solver_utils.forward_warp import forward_warp, inverse_warp
    H, W = 576, 1024
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = (2 + 0.5 * np.sin(xs / 97) + 0.3 * np.cos(ys / 53)).astype(np.float32)
    K = np.array([[800, 0, W / 2], [0, 800, H / 2], [0, 0, 1]], np.float32)
    T1 = np.eye(4, dtype=np.float32)
    T2 = np.eye(4, dtype=np.float32)
    T2[0, 3], T2[2, 3] = 0.05, 0.02
    rgb = np.random.default_rng(0).random((3, H, W), dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def host_s(fn):                       # median of three runs
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]

    def dev_us(fn, n=5):
        fn()
        torch.cuda.synchronize()
        with L.kernel_trace() as tr:
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        return 1e3 * sum(v[1] for v in tr.result.values()) / n

    out = {}
    img, d, k, p1, p2 = t(rgb), t(depth), t(K), t(T1), t(T2)
    frame = (rgb.transpose(1, 2, 0) * 255).astype(np.float64)
    f64 = lambda a: a.astype(np.float64)
    out["W1_forward_warp_576x1024"] = dict(
        cpu_port_s=round(host_s(lambda: WO.forward_warp(frame, None, f64(depth), f64(T1), f64(T2), f64(K), None)), 3),
        gpu_device_us=round(dev_us(lambda: forward_warp(frame, None, f64(depth), f64(T1), f64(T2), f64(K), None)), 1))
    out["W2_inverse_warp_576x1024"] = dict(
        cpu_port_s=round(host_s(lambda: WO.inverse_warp(rgb, depth, depth, T1, T2, K, 20)), 3),
        gpu_device_us=round(dev_us(lambda: inverse_warp(img, d[None], d[None], p1, p2, k, bandwidth=20)), 1))
    out["C1_consistency_check_576x1024"] = dict(
        cpu_port_s=round(host_s(lambda: WO.consistency_check_with_depth(depth, T2, K, depth, T1, K)), 3),
        gpu_device_us=round(dev_us(lambda: consistency_check_with_depth(d, p2, k, d, p1, k)), 1))
    c = GI.sched_case("full_f16")
    sch = EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG)
    sch.set_timesteps(100)
    sig, i = sch.sigmas.numpy(), c["step_i"]
    g = {n: t(c[n]) for n in ("model_output", "sample", "temp_cond", "mask")}
    lam = torch.from_numpy(c["lambda_ts"])
    ts = sch.timesteps[i]
    for name, grad in (("S2_step_interp_grad", True), ("S2_step_interp", False)):
        out[name + "_25x4x72x128"] = dict(
            cpu_port_s=round(host_s(lambda: SO.step_interp(c["model_output"], c["sample"], c["temp_cond"], c["mask"], c["lambda_ts"][i], sig, i,
                                                           lr=0.02, compute_grad=grad)), 3),
            gpu_device_us=round(dev_us(lambda: sch.step_interp(g["model_output"], ts, g["sample"], g["temp_cond"], g["mask"], lam, step_i=i,
                                                               lr=0.02, compute_grad=grad)), 1))
    out["S3_step_interp_prob_uncertain_25x4x72x128"] = dict(
        cpu_port_s=round(host_s(lambda: SO.step_interp_prob_uncertain(c["model_output"], c["sample"], c["temp_cond"], c["mask"],
                                                                      c["lambda_ts"][i], sig, i)), 3),
        gpu_device_us=round(dev_us(lambda: sch.step_interp_prob_uncertain(g["model_output"], ts, g["sample"], g["temp_cond"], g["mask"], lam,
                                                                          step_i=i)), 1))
    out["note"] = ("cpu_port_s: oracle/warp_oracle.py / oracle/scheduler_oracle.py (numpy, median of 3 runs, host cores as numpy uses them); "
                   "gpu_device_us: sum of kernel time of the HIP path on the same inputs.  Reference-run timings of the same functions "
                   "in the build container (8 cores): BASELINE.md section 3")
    return out


def cpu_baseline(args, with_unet: bool):
    """Composite CPU figure in the headline's unit: raster iterations per second of the same block
    (raster_iters raster iterations + one SVD unit), from the two bounded oracle samples."""
    ras = cpu_baseline_raster(args)
    t_iter = 1.0 / ras["value"]
    if not with_unet:
        return ras
    t_unit, note = cpu_baseline_unet(args)
    value = args.raster_iters / (args.raster_iters * t_iter + t_unit)
    return dict(value=value, unit="iters/s", cores=ras["cores"], kind="port",
                sample=ras["sample"].replace("; raster iterations only (no CPU UNet: see DESIGN.md)", "")
                       + f" -> {t_iter:.1f} s per raster iteration; {note} -> {t_unit:.0f} s per SVD unit; "
                         f"value = {args.raster_iters} / ({args.raster_iters} x raster + unit)")


def other_rooflines(kern, alg, loop_b, units):
    """Roofline entries of the other kernels of the path next to the dominant one (same HIP-event trace): the blend kernels
    against the vector-instruction ISSUE roof that binds them (executed VALU wave-instructions per launch / measured launch time
    against 1024 SIMDs x 2.4 GHz / 4 cycles), SURVEY 8d's byte model beside it, and the spatial attention against the MFMA peak."""
    out = {}
    from syn3r_amd.pipeline.svd_step import _pmc_traffic
    for name in ("k_render_bwd", "k_render"):
        if name in kern and name in alg:
            calls, ms = kern[name]
            avg_s = ms / calls / 1e3
            # the roof that binds them: vector-instruction ISSUE.  Executed wave-instructions per launch come
            # from the SQ_INSTS_VALU pass of the committed PMC summary (this build's if it has one), the launch time is measured here.
            insts, src = _pmc_traffic(name, field="valu_insts_per_launch")
            stale = insts is None       # no PMC summary of THIS build's sources: an earlier build's count, said so in the line
            if stale:
                insts, src = VALU_INSTS_FALLBACK[name]
            ginst = insts / avg_s / 1e9
            ach = alg[name] / avg_s / 1e9
            out[name] = dict(bound="valu_issue", achieved=round(ginst, 1), peak=round(VALU_ISSUE_PEAK_GINST, 1), unit="G wave-instructions/s",
                             frac=round(ginst / VALU_ISSUE_PEAK_GINST, 4), stale=stale,
                             frac_of_measured_issue_rate=round(ginst / VALU_ISSUE_MEASURED_GINST, 4),
                             measured_issue_rate=dict(value=round(VALU_ISSUE_MEASURED_GINST, 1), unit="G wave-instructions/s",
                                                      source="profiles/r06/valu_rate.txt: v_fma_f32, 2 wavefronts per SIMD, 2.02 ns per instruction and SIMD"),
                             valu_insts_per_launch=int(insts), valu_insts_source=src,
                             avg_ms=round(ms / calls, 4), calls=calls,
                             hbm_byte_model=dict(note="SURVEY 8d byte model: how far the traffic is from mattering, not the binding roof",
                                                 achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                                                 algorithmic_bytes=alg[name]))
    if loop_b is not None and "k_attn_spatial" in kern and loop_b.flops_per_unit:
        calls, ms = kern["k_attn_spatial"]
        ach = loop_b.flops_per_unit["attn"] * units / (ms / 1e3) / 1e12
        out["k_attn_spatial"] = dict(bound="mfma", achieved=round(ach, 1), peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s",
                                     frac=round(ach / MFMA_F16_PEAK_TFLOPS, 4), avg_ms=round(ms / calls, 4), calls=calls)
    return out


def sub_benchmarks(args, dev, loop_a, loop_b, log):
    """Timed here, by the same process, AFTER the headline's timed region (rank 0 of the 1-GPU run): the configurations
    the reference's scripts actually run next to BASELINE config 2's — the Post unit at F = 25
    (bash_scripts/batch_llff_train.sh:39), the Replace unit at F = 25 (batch_dtu_train.sh:42), a complete trainer
    iteration (L1 + SSIM + Adam), and the fused inverse warp at 576x1024 with its SURVEY 8d roofline."""
    from syn3r_amd import _lib as L
    out = {}

    def wall_ms(fn, n):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    if loop_b is not None:
        from syn3r_amd.pipeline.svd_step import SvdStepBench
        out["svd_post_unit_f14_ms"] = round(wall_ms(loop_b.step_pass_post, 3), 2)
        b25 = SvdStepBench(25, dev, seed=args.seed, unet=loop_b.unet)
        # (step, pass) units at F = 25 the way the pipelines run them: both passes of a step stacked into one UNet launch
        # sequence (merge_passes) = 2 units per call; the pass-by-pass figures beside them
        out["svd_replace_unit_f25_ms"] = round(wall_ms(lambda: b25.step_both("replace"), 3) / 2, 2)
        out["svd_post_unit_f25_ms"] = round(wall_ms(lambda: b25.step_both("post"), 3) / 2, 2)
        out["svd_unit_f25_pass_by_pass_ms"] = {"replace": round(wall_ms(b25.step_pass, 3), 2), "post": round(wall_ms(b25.step_pass_post, 3), 2)}
        del b25
        torch.cuda.empty_cache()
        log("sub-benchmarks: SVD units done")
    # HOT LOOP A alone: the headline's raster iteration (render, fused L1, backward) back to back, no SVD unit in between
    it_ms = wall_ms(loop_a.iteration, 100)
    out["raster_only_iters_per_s"] = round(1e3 / it_ms, 1)
    out["raster_only_iteration_us"] = round(1e3 * it_ms, 1)
    out["raster_full_iteration_per_s"] = round(loop_a.full_iteration_rate(50), 1)
    out["raster_full_iteration_note"] = "render + (0.8 L1 + 0.2 (1-SSIM)) + backward + fused Adam on 5 parameter groups, no host sync"
    if loop_b is not None and not args.no_end_to_end:
        # MEASURED end to end (syn3r_amd/measure.py): one full-size svd_render per variant (26 VAE encodes, 100 steps x 2 passes
        # at F = 25, chunked decode: SVD_2pass_prob_uncertain_post.py:544-848) and a scaled schedule through DiffusionGS.run
        # (500 + 500 trainer iterations around ONE densified view pair, model/diffusionGS.py:1668-1697)
        import tempfile
        from syn3r_amd import measure as M
        comps = M.full_size_components(dev, unet=loop_b.unet, seed=args.seed)
        out["denoise_host_gap"] = {v: M.measure_denoise_gap(comps, v, dev) for v in ("replace", "post")}
        rep = M.measure_svd_render(comps, "replace", dev, steps=args.e2e_steps)
        log(f"sub-benchmarks: measured svd_render (replace): {rep['wall_s']} s")
        with tempfile.TemporaryDirectory() as tmp:
            sched = M.measure_schedule(comps, dev, tmp, variant="post", N=args.gaussians, H=args.height, W=args.width,
                                       iterations=args.e2e_iterations, steps=args.e2e_steps, seed=args.seed)
        log(f"sub-benchmarks: measured schedule (post): {sched['wall_s']} s")
        out["svd_render_f25_s"] = {"measured": True, "steps": args.e2e_steps, "replace": rep["wall_s"], "post": sched["svd_render_s"],
                                   "replace_stages": {k: rep[k] for k in ("vae_encode_s", "denoise_s", "vae_decode_s", "denoise_ms_per_step_pass")},
                                   "note": "wall-clock of StableVideoDiffusionPipeline.__call__ with the HIP VAE and UNet (seeded weights); "
                                           "'post' is the call DiffusionGS.svd_render made inside schedule_scaled"}
        out["schedule_scaled"] = sched
        # the fern-like schedule 10 k + 2 x (3 view pairs + 10 k), from the MEASURED stage rates of the scaled run
        # (trainer iterations/s inside training()/finetune(), one whole view pair incl. its svd_render) and, beside it, the
        # round-2 derivation from the isolated unit rates
        it_s = 30000.0 / sched["trainer_iters_per_s"]
        out["fern_like_schedule_s"] = {
            "from_measured_stages": {"raster_30k_iterations": round(it_s, 1), "post": round(it_s + 6 * sched["view_pair_s"], 1),
                                     "replace": round(it_s + 6 * (sched["view_pair_s"] - sched["svd_render_s"] + rep["wall_s"]), 1)},
            "derived_from_unit_rates": {"formula": "30000 / raster_full_iteration_per_s + 6 x 200 x unit_f25_ms",
                                        "replace": round(30000.0 / out["raster_full_iteration_per_s"] + 1.2 * out["svd_replace_unit_f25_ms"], 1),
                                        "post": round(30000.0 / out["raster_full_iteration_per_s"] + 1.2 * out["svd_post_unit_f25_ms"], 1)}}
        del comps
        torch.cuda.empty_cache()
    # fused inverse warp + reprojection consistency (W2 + C1) at the reference's 576x1024 working size
    from syn3r_amd.solver_utils.forward_warp import inverse_warp
    H, W = 576, 1024
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    depth = torch.from_numpy((2 + 0.5 * np.sin(xs / 97) + 0.3 * np.cos(ys / 53)).astype(np.float32)).to(dev)
    K = torch.tensor([[800, 0, W / 2], [0, 800, H / 2], [0, 0, 1]], dtype=torch.float32, device=dev)
    T1 = torch.eye(4, device=dev)
    T2 = torch.eye(4, device=dev); T2[0, 3], T2[2, 3] = 0.05, 0.02
    img = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)
    fn = lambda: inverse_warp(img, depth[None], depth[None], T1, T2, K, bandwidth=20)
    fn(); torch.cuda.synchronize()
    with L.kernel_trace() as tr:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    us = 1e3 * sum(v[1] for v in tr.result.values()) / 20
    gbs = 49.0 * H * W / (us * 1e-6) / 1e9
    out["inverse_warp_576x1024"] = {"device_us": round(us, 1), "kernels": {k: round(1e3 * v[1] / 20, 1) for k, v in tr.result.items()},
                                    "roofline": {"bound": "hbm", "algorithmic_bytes": 49 * H * W, "achieved": round(gbs, 1),
                                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                 "note": "29 MB per call: two launches of ~13 us each, latency- not bandwidth-limited"}}
    return out


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (never an exec: this process may
    not have touched the GPU yet, but the box refuses exec from GPU processes anyway) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)     # stderr passes through
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    for l in r.stdout.splitlines():
        if l.strip() and not l.strip().startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if (r.returncode or lines) else 1


def main():
    args = parse()
    # BEFORE anything touches the GPU: --gpus N without a torchrun environment means "start the N ranks yourself"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the SYN3R hot path has no CPU fallback)")
    # rehearsal knobs (not used by the driver): all ranks on one GPU and/or gloo instead of RCCL
    if os.environ.get("SYN3R_BENCH_SINGLE_DEVICE") == "1":
        os.environ["LOCAL_RANK"] = "0"
    from syn3r_amd import dist as D
    rank, world, local = D.init(os.environ.get("SYN3R_BENCH_BACKEND"))     # one process per GPU; "nccl" is RCCL
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launch environment has WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = torch.distributed if world > 1 else None
    from syn3r_amd import _lib as L
    L.load()

    # one independent scene per rank (scene-parallel): different seed per rank
    args.seed = args.seed + rank
    loop_a = RasterLoop(args, dev)
    loop_b = None
    if args.svd == "on":
        try:
            from syn3r_amd.pipeline.svd_step import SvdStepBench
            loop_b = SvdStepBench(args.frames, dev, seed=args.seed)
        except ImportError:
            loop_b = None

    def step():
        for _ in range(args.raster_iters):
            loop_a.iteration()
        if loop_b is not None:
            loop_b.step_pass()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    log("setup done; warmup")
    for _ in range(args.warmup):
        step()
    if loop_b is not None:
        loop_b.count_flops()
    barrier()
    # start/stop events only on the kernels that can be the dominant one (a timed launch costs microseconds):
    # the contraction family and the spatial attention when the SVD unit is part of the step, else the blend kernels
    only = "k_gemm,k_attn_spatial,k_render" if loop_b is not None else "k_render"
    # ... and only during the last --trace-steps of the timed steps: still inside the timed region, for a fraction
    # of the events' cost (2-3 % of a step when every launch of every step is timed)
    traced = 0 if args.no_kernel_trace else max(1, min(args.trace_steps, args.steps))
    t0 = time.perf_counter()
    for _ in range(args.steps - traced):
        step()
    with L.kernel_trace(only=only if traced else "(none)") as tr:
        for _ in range(traced):
            step()
        from syn3r_amd import raster as _r
        truncated = _r.flush_pair_checks()         # every render's pair list was complete (raises otherwise)
        barrier()
        dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f} s")
    # ONE collective at the end (north_star): the fixed per-scene record of syn3r_amd/dist.py; the job's time is
    # the MAX of the ranks' wall-clocks, read from the gathered records
    psnr, ssim = loop_a.metrics()
    rec = [float(rank), psnr, ssim, float("nan"), args.raster_iters * args.steps / dt, (args.steps / dt) if loop_b else 0.0, dt, float(truncated), 1.0]
    allrec = D.gather_records(rec)
    dt_max = float(allrec[:, D.RECORD_FIELDS.index("wall_s")].max())

    if rank == 0:
        iters = args.raster_iters * args.steps * world
        value = iters / dt_max
        P = loop_a.pairs()
        alg = raster_algorithmic_bytes(args.gaussians, P, args.height, args.width)
        kern = {k: v for k, v in tr.result.items()}
        roof = None
        if loop_b is not None:
            roof = loop_b.roofline(kern, max(traced, 1))
        if roof is None and kern:
            name = max(kern, key=lambda k: kern[k][1])
            calls, ms = kern[name]
            avg_s = ms / calls / 1e3
            if name in alg:
                ach = alg[name] / avg_s / 1e9
                roof = dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, avg_ms=round(ms / calls, 4), calls=calls,
                            algorithmic_bytes=alg[name])
            else:
                roof = dict(bound="hbm", kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None,
                            traffic=None, avg_ms=round(ms / calls, 4), calls=calls)
        out = {
            "metric": "llff_3view_train_loop_iters_per_sec",
            "value": round(value, 3),
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 raster / f16 UNet (f32 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": f"LLFF fern-like 3-view block: {args.raster_iters} raster fwd+bwd iters ({args.loss} loss) "
                            f"({args.gaussians} Gaussians, {args.width}x{args.height}, SH3, P={P} pairs) + "
                            + (f"1 SVD (step,pass): CFG UNet fwd [2,{args.frames},8,72,128] f16 + fused Euler step"
                               if loop_b is not None else "SVD pass NOT included (UNet not built yet)"),
                "raster_iters_per_step": args.raster_iters,
                "svd_units_per_step": 1 if loop_b is not None else 0,
                "frames": args.frames,
                "parallelism": f"scene-parallel x{world}",
            },
            "roofline": dict(roof, traced_steps=traced) if roof else roof,
            "roofline_other": other_rooflines(kern, alg, loop_b, max(traced, 1)),
            "kernels_ms": {k: [v[0], round(v[1], 3)] for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])[:12]},
            "record_fields": list(D.RECORD_FIELDS),
            "per_rank": [[None if x != x else round(float(x), 3) for x in r.tolist()] for r in allrec],
        }
        if loop_b is not None and world == 1:
            # the other reading of "SVD_1pass" in the fern-like schedule: 1 200 (step, pass) units per 30 000 iterations = 25 : 1
            # (bench.py header: the headline counts 600 one-pass units, 50 : 1).  Timed here, same process, same kernels.
            half = max(1, args.raster_iters // 2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                for _ in range(half):
                    loop_a.iteration()
                loop_b.step_pass()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            out["value_two_pass_mix"] = {"value": round(2 * half / dt2, 3), "unit": "iters/s", "raster_iters_per_svd_unit": half,
                                         "ms_per_step": round(1e3 * dt2 / 2, 3), "steps": 2,
                                         "note": "same step with one SVD unit per %d raster iterations (the two-pass reading of the schedule)" % half}
        if not args.no_sub_benchmarks and world == 1:
            log("sub-benchmarks (Post / Replace units at F = 25, full trainer iteration, inverse warp) ...")
            out["sub_benchmarks"] = sub_benchmarks(args, dev, loop_a, loop_b, log)
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 of the single-GPU run only
            log("cpu baseline (oracle on the host cores, bounded sample) ...")
            out["cpu_baseline"] = cpu_baseline(args, loop_b is not None)
            out["cpu_baseline"]["sample_note"] = ("bounded samples: one oracle run of the raster leg (1/4 of the pixels) and of the UNet leg (1/14 of "
                                                  "the tokens), extrapolated linearly - the UNet leg's token-linear scaling UNDER-counts the spatial attention, "
                                                  "whose work is quadratic in the tokens of a frame (the sample keeps whole 72x64 frames, half of 72x128: its "
                                                  "attention is 1/4 per frame, scaled x2), so the CPU time per unit is a lower bound; geometry / scheduler legs: "
                                                  "median of 3; kind 'port' because reference Python does not travel to the GPU box")
            out["cpu_baseline"]["geometry_and_scheduler"] = cpu_baseline_geometry_scheduler(dev)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
