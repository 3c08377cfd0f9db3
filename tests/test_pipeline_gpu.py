"""SVD two-pass pipelines (P1 "Post", P2 "Replace"): the HIP pipeline's loop logic vs latents produced by the
REFERENCE pipeline classes' own __call__ (run on the CPU with the mock CLIP / VAE / UNet of
oracle/pipeline_mocks.py; tests/golden/pipeline_mock.npz).  The scheduler steps inside run on the HIP kernels."""
import numpy as np
import pytest
import torch

from oracle import pipeline_mocks as PM

pytestmark = pytest.mark.gpu


def make_pipe(variant, dev):
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    return StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), PM.MockUNet().to(dev),
                                        EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG), variant=variant, device=dev)


def run(variant, dev, **kw):
    inp = PM.pipeline_inputs(seed=0)
    pipe = make_pipe(variant, dev)
    res = pipe([im.to(dev) for im in inp["image"]], temp_cond=[t.to(dev) for t in inp["temp_cond"]],
               mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8,
               num_inference_steps=3, latent_num=1, latents=inp["latents"].clone(), output_type="latent",
               dtype=torch.float32, aug_noise=inp["noise"], **kw)
    return res.frames


@pytest.mark.parametrize("variant", ["post", "replace"])
def test_pipeline_latents_match_reference(variant, gpu, golden_dir):
    g = np.load(golden_dir / "pipeline_mock.npz")[variant]
    lat = run(variant, gpu)
    assert lat.shape == (1, 25, 4, 72, 128) and lat.dtype == torch.float32
    a = lat.cpu().numpy()[..., ::3, ::3]
    scale = np.abs(g).max()
    # three denoise steps at sigma 700 -> ~1: selection ties with the quantile cut-off may flip single latents
    bad = np.abs(a - g) > 2e-3 * scale
    assert bad.mean() < 2e-3, (bad.mean(), np.abs(a - g).max(), scale)


@pytest.mark.parametrize("variant", ["post", "replace"])
def test_one_pass_matches_reference_forward_branch(variant, gpu, golden_dir):
    """BASELINE configs[1] "SVD_1pass" (`one_pass=True`): the forward-in-time pass only.  Golden = the REFERENCE
    two-pass classes' own __call__ with the blend weight linspace(1,0,F) forced to ones (oracle/gen_golden.py
    pipeline_one_pass; tests/golden/pipeline_one_pass.npz), i.e. latents = forward branch at every step."""
    g = np.load(golden_dir / "pipeline_one_pass.npz")[variant]
    lat = run(variant, gpu, one_pass=True)
    assert lat.shape == (1, 25, 4, 72, 128)
    a = lat.cpu().numpy()[..., ::3, ::3]
    scale = np.abs(g).max()
    bad = np.abs(a - g) > 2e-3 * scale
    assert bad.mean() < 2e-3, (bad.mean(), np.abs(a - g).max(), scale)
    two = np.load(golden_dir / "pipeline_mock.npz")[variant]
    assert np.abs(g - two).max() > 1e-2 * scale          # and it is not the two-pass result


def test_pipeline_argument_errors(gpu):
    pipe = make_pipe("post", gpu)
    inp = PM.pipeline_inputs(seed=0)
    with pytest.raises(ValueError):
        pipe(inp["image"], temp_cond=inp["temp_cond"][:5], mask=inp["mask"], lambda_ts=inp["lambda_ts"], num_frames=25)
    with pytest.raises(NotImplementedError):
        make_pipe("1pass_prob", gpu)


@pytest.mark.parametrize("variant", ["replace", "post"])
def test_pipeline_with_hip_unet_matches_reference_pipeline_with_reference_unet(variant, gpu, golden_dir):
    """End to end: the HIP pipeline driving the HIP UNet (fp16) vs the REFERENCE pipeline class driving the
    REFERENCE UNetSpatioTemporalConditionModel (CPU fp32) on identical seeded weights and inputs
    (tests/golden/pipeline_unet.npz, oracle/gen_golden.py pipeline_unet): CFG batch, the Post variant's guidance
    tiles and gradient step, time flips and the forward/backward blend, through two denoise steps."""
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    g = np.load(golden_dir / "pipeline_unet.npz")[variant]
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
    pipe = StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant=variant, device=gpu)
    inp = PM.pipeline_inputs(seed=1)
    lat = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]],
               mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8,
               num_inference_steps=2, latent_num=1, latents=inp["latents"].clone(), output_type="latent",
               dtype=torch.float16, aug_noise=inp["noise"]).frames
    assert lat.shape == (1, 25, 4, 72, 128)
    a = lat.float().cpu().numpy()[..., ::3, ::3]
    scale = np.abs(g).max()
    err = np.abs(a - g)
    # fp16 UNet against the fp32 reference over two Euler steps from sigma = 700; the quantile selection of the
    # guidance / replacement steps may flip single latents next to the cut-off
    # (measured: mean 2-4e-4, max 2.5e-3 of the latent scale, tools/pipeline_parity.py)
    assert err.mean() < 1e-3 * scale, (err.mean(), scale)
    assert (err > 1e-2 * scale).mean() < 1e-4, ((err > 1e-2 * scale).mean(), err.max(), scale)


def test_pipeline_with_hip_unet_and_vae_matches_reference_frames(gpu, golden_dir):
    """The decoded frames: HIP pipeline + HIP UNet + HIP temporal-decoder VAE vs the reference pipeline driving the
    reference UNet and the reference AutoencoderKLTemporalDecoder (CPU fp32, reduced configurations, identical seeded
    weights; tests/golden/pipeline_unet_vae.npz): the condition-image encodes (scaled, with the shared augmentation
    noise), the latent scaling and the chunked temporal decode, end to end ('replace' variant, two steps)."""
    from oracle import unet_weights as UW
    from oracle import vae_weights as VW
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    from syn3r_amd.vae import AutoencoderKLTemporalDecoder
    g = np.load(golden_dir / "pipeline_unet_vae.npz")["frames"]
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
    vae = AutoencoderKLTemporalDecoder(**VW.PIPELINE_VAE_CONFIG)
    vae.load_state_dict(UW.make_state_dict(vae.parameter_shapes(), seed=11), gpu)
    pipe = StableVideoDiffusionPipeline(vae, PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant="replace", device=gpu)
    inp = PM.pipeline_inputs(seed=2)
    frames = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]],
                  mask=inp["mask"].clone(), lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8,
                  num_inference_steps=2, latent_num=1, latents=inp["latents"].clone(), output_type="np",
                  dtype=torch.float16, aug_noise=inp["noise"]).frames[0]
    a = np.asarray(frames, dtype=np.float32)
    assert a.shape == (25, 576, 1024, 3)
    err = np.abs(a[:, ::16, ::16] - g)
    # frames are in [0, 1]; fp16 encoder / UNet / decoder against the fp32 reference
    # (measured: mean 6.4e-4, max 6.1e-3, tools/pipeline_parity.py)
    assert err.mean() < 2e-3, err.mean()
    assert (err > 2e-2).mean() < 1e-4, ((err > 2e-2).mean(), err.max())


@pytest.mark.parametrize("variant", ["replace", "post"])
def test_merged_passes_equal_separate_passes(variant, gpu):
    """`merge_passes`: the forward- and backward-in-time passes of a step stacked into one UNet launch sequence (B = 4 CFG
    calls; for Post also the guidance tiles of both passes, with one shared or two distinct unconditional contexts) give
    the latents of the pass-by-pass loop."""
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
    inp = PM.pipeline_inputs(seed=4)

    outs = {}
    for distinct in (False, True):
        for merge in (False, True):
            pipe = StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                                variant=variant, device=gpu)
            if distinct:                  # start / end passes with DIFFERENT unconditional embeddings (tile contexts per pass)
                enc = pipe._encode_image
                count = [0]

                def encode(image, do_cfg, enc=enc, count=count):
                    e = enc(image, do_cfg).clone()
                    count[0] += 1
                    e[0] += 0.05 * count[0]
                    return e
                pipe._encode_image = encode
            outs[(distinct, merge)] = pipe(
                [im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]], mask=inp["mask"].clone(),
                lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8, num_inference_steps=2, latent_num=1,
                latents=inp["latents"].clone(), output_type="latent", dtype=torch.float16, aug_noise=inp["noise"],
                merge_passes=merge).frames.float()
        a, b = outs[(distinct, False)], outs[(distinct, True)]
        scale = float(a.abs().max())
        err = (a - b).abs()
        assert float(err.mean()) < 5e-4 * scale and float((err > 1e-2 * scale).float().mean()) < 1e-4, (float(err.mean()), float(err.max()), scale)
    assert not torch.allclose(outs[(False, True)], outs[(True, True)], atol=1e-3 * scale)       # the patched contexts did matter


def test_full_size_replace_pipeline_matches_reference(gpu, golden_dir):
    """FULL SIZE: the HIP pipeline + the HIP UNet in the SVD-XT configuration (1.52 B parameters, fp16) against the REFERENCE
    `SVD_2pass_prob_uncertain` pipeline class driving the REFERENCE UNet (CPU fp32) on identical seeded weights and inputs
    (tests/golden/pipeline_unet_full.npz, `oracle/gen_golden.py pipeline_full`): one denoising step, both passes — stacked
    into one B = 4 call here — at [*,25,8,72,128], soft replacement, time flip and blend.  (The Post variant at this size:
    test_full_size_post_pipeline_matches_reference below.)"""
    path = golden_dir / "pipeline_unet_full.npz"
    if not path.exists():
        pytest.skip("pipeline_unet_full.npz not generated (oracle/gen_golden.py pipeline_full, ~30 min of CPU)")
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    g = np.load(path)
    unet = UNetSpatioTemporalConditionModel()
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=5), gpu)
    pipe = StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant="replace", device=gpu)
    inp = PM.pipeline_inputs(seed=6)
    lat = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]], mask=inp["mask"].clone(),
               lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8, num_inference_steps=1, latent_num=1,
               latents=inp["latents"].clone(), output_type="latent", dtype=torch.float16, aug_noise=inp["noise"]).frames
    a = lat.float().cpu().numpy()
    ref = g["replace"]
    scale = np.abs(ref).max()
    err = np.abs(a[..., ::2, ::2] - ref)
    assert err.mean() < 1e-3 * scale, (err.mean(), scale)
    assert (err > 1e-2 * scale).mean() < 1e-4, ((err > 1e-2 * scale).mean(), err.max(), scale)
    assert abs(float(np.abs(a).mean()) - float(g["mean_abs"])) < 2e-3 * scale


def test_full_size_post_pipeline_matches_reference(gpu, golden_dir):
    """FULL SIZE, the variant every LLFF / DL3DV script runs (SVD_2pass_prob_uncertain_post.py:725-800): the HIP pipeline + the
    HIP UNet in the SVD-XT configuration against the REFERENCE Post pipeline class and scheduler driving the REFERENCE UNet
    (CPU fp32, 1.52 B seeded parameters; tests/golden/pipeline_unet_full_post.npz, `oracle/gen_golden.py pipeline_full_post`):
    one denoising step = both passes, each the four overlapping guidance tiles [1,25,8,40|48,72] (closed-form gradient here,
    autograd of the scheduler's loss there), the stitch, and the CFG forward at [2,25,8,72,128].  The fixture's reference UNet
    ran behind a no-grad wrapper (the gradient the pipeline consumes never passes through the UNet: `gen_golden.py
    nograd_check` reproduces the with-graph reduced-width fixture to fp32 rounding, 7e-5 relative)."""
    path = golden_dir / "pipeline_unet_full_post.npz"
    if not path.exists():
        pytest.skip("pipeline_unet_full_post.npz not generated (oracle/gen_golden.py pipeline_full_post, ~25 min of CPU)")
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_2pass import StableVideoDiffusionPipeline
    from syn3r_amd.schedulers.scheduling_euler_discrete import SVD_XT_SCHEDULER_CONFIG, EulerDiscreteScheduler
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    g = np.load(path)
    unet = UNetSpatioTemporalConditionModel()
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=5), gpu)
    pipe = StableVideoDiffusionPipeline(PM.MockVAE(), PM.MockImageEncoder(), unet, EulerDiscreteScheduler(**SVD_XT_SCHEDULER_CONFIG),
                                        variant="post", device=gpu)
    inp = PM.pipeline_inputs(seed=6)
    lat = pipe([im.to(gpu) for im in inp["image"]], temp_cond=[t.to(gpu) for t in inp["temp_cond"]], mask=inp["mask"].clone(),
               lambda_ts=inp["lambda_ts"], num_frames=25, decode_chunk_size=8, num_inference_steps=1, latent_num=1,
               latents=inp["latents"].clone(), output_type="latent", dtype=torch.float16, aug_noise=inp["noise"]).frames
    a = lat.float().cpu().numpy()
    ref = g["post"]
    scale = np.abs(ref).max()
    err = np.abs(a[..., ::2, ::2] - ref)
    assert err.mean() < 1e-3 * scale, (err.mean(), scale)
    assert (err > 1e-2 * scale).mean() < 1e-4, ((err > 1e-2 * scale).mean(), err.max(), scale)
    assert abs(float(np.abs(a).mean()) - float(g["mean_abs"])) < 2e-3 * scale


def test_replace_passes_on_two_streams_equal_the_pass_after_pass_order(gpu):
    """`_streamed_replace`: from the second step of a shape on, the two passes of a Replace step run on two HIP streams (their
    kernels fill each other's partly empty last rounds).  Same kernels on the same operands as the pass-after-pass order:
    bit-identical to it; the first call of a shape issues the same two sequences on one stream."""
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_step import SvdStepBench
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
    b = SvdStepBench(25, gpu, h=16, w=24, unet=unet)
    b.step_both("replace")                                   # first call of the shape: the same sequences on ONE stream (creates the shared caches)
    st = b._both_replace
    pipe = st["pipe"]
    assert pipe.two_streams and ("replace", (1, 25, 4, 16, 24)) in pipe._streams_warm
    i, t = 7, b.sch.timesteps[7]
    lat = (b.latents, b.latents.flip(dims=[1]))
    assert pipe._side is None                                # ... and did not touch the side streams
    streamed = pipe._streamed_replace(i, t, lat, st["img4"], st["ehs4"], st["added4"], st["ops2"], True)
    torch.cuda.synchronize()
    assert pipe._side is not None                            # the side streams were used
    seq = []
    for k in range(2):
        cond, mask, lam, _ = st["ops2"][k]
        seq.append(pipe._pass_replace(i, t, lat[k], st["img4"][2 * k:2 * k + 2], st["ehs4"][2 * k:2 * k + 2], st["added4"][2 * k:2 * k + 2],
                                      cond, mask, lam, True))
    for a, c in zip(streamed, seq):
        assert torch.equal(a, c)
    pipe.two_streams = False
    stacked = pipe._streamed_replace(i, t, lat, st["img4"], st["ehs4"], st["added4"], st["ops2"], True)
    for a, c in zip(streamed, stacked):                      # the stacked order: the same latents up to fp16 rounding of regrouped tiles
        scale = float(c.float().abs().max())
        assert float((a.float() - c.float()).abs().mean()) < 5e-4 * scale


def test_post_step_on_three_streams_equals_the_one_stream_orders(gpu):
    """`_merged_post` from the second step of a shape on: each pass's CFG forward on its own stream, the stacked guidance-tile
    forwards beside them on the current one.  The CFG halves are the pass-after-pass arithmetic, the tiles the stacked one: the
    latents equal BOTH one-stream orders within the tolerance those two have against each other, and a second streamed call
    reproduces the first bit for bit (no race between the sequences)."""
    from oracle import unet_weights as UW
    from syn3r_amd.pipeline.svd_step import SvdStepBench
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    unet = UNetSpatioTemporalConditionModel(**UW.PIPELINE_CONFIG)
    unet.load_state_dict(UW.make_state_dict(unet.parameter_shapes(), seed=3), gpu)
    b = SvdStepBench(25, gpu, h=72, w=128, unet=unet)          # (the guidance tiles need a 9 x 16-divisible grid with 8-divisible tiles)
    b.step_both("post")                                      # first call of the shape: the same sequences on one stream (creates the shared caches)
    st = b._both_post
    pipe = st["pipe"]
    i, t = 7, b.sch.timesteps[7]
    lat = (b.latents, b.latents.flip(dims=[1]))
    args = (i, t, lat, st["img4"], st["ehs4"], st["added4"], st["ops2"], True, st["tile_ctx"])
    assert pipe.two_streams
    s1 = pipe._merged_post(*args)
    s2 = pipe._merged_post(*args)
    torch.cuda.synchronize()
    assert pipe._side is not None
    for a, c in zip(s1, s2):
        assert torch.equal(a, c)
    pipe.two_streams = False
    stacked = pipe._merged_post(*args)
    seq = []
    for k in range(2):
        cond, mask, lam, tops = st["ops2"][k]
        seq.append(pipe._pass_post(i, t, lat[k], st["img4"][2 * k:2 * k + 2], st["ehs4"][2 * k:2 * k + 2], st["added4"][2 * k:2 * k + 2],
                                   cond, mask, lam, True, tops))
    for ref in (stacked, seq):
        for a, c in zip(s1, ref):
            scale = float(c.float().abs().max())
            err = (a.float() - c.float()).abs()
            assert float(err.mean()) < 5e-4 * scale and float((err > 1e-2 * scale).float().mean()) < 1e-4, (float(err.mean()), float(err.max()), scale)
