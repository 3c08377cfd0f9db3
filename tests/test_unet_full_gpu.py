"""The SVD-XT-sized UNet (default configuration: 320/640/1280/1280 channels, 5/10/20/20 heads, 1.52 B parameters)
with the DEFAULT kernel dispatch — the contraction / attention variants bench.py times — against

  * tests/golden/unet_full_*.npz: outputs of the REFERENCE `UNetSpatioTemporalConditionModel()` (CPU fp32,
    name-keyed seeded weights, `oracle/gen_golden.py unet_full`) at the shapes the benchmark and the pipelines
    launch: [2,14,8,72,128] (bench unit), [1,25,8,40,72] / [1,25,8,48,72] (Post guidance tiles), [2,25,8,72,128];
  * oracle/unet_oracle.py (pinned by unet_small.npz) at a small shape, run live.

Tolerance: fp16 storage / fp32 accumulate through ~300 layers against fp32.  Measured on MI355X (round 5, recorded by
tests/conftest.py:record_measurement, gpurun_out/test_measurements.jsonl): max 1.23-1.97e-3, mean 2.0-3.3e-4 of the output scale
over the four shapes and the live oracle shape; the bar is 4e-3 max / 6.6e-4 mean = 2x the largest measurement (VERDICT r04:
a 4x regression of the accumulated error used to pass)."""
import numpy as np
import pytest
import torch

from oracle import unet_weights as UW

pytestmark = pytest.mark.gpu

FULL_SEED = 5      # oracle/gen_golden.py gen_unet_full
MAX_BAR, MEAN_BAR = 4e-3, 6.6e-4


@pytest.fixture(scope="module")
def full_unet(gpu):
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel()
    sd = UW.make_state_dict(model.parameter_shapes(), seed=FULL_SEED)
    model.load_state_dict(sd, gpu)
    return model, sd


def _compare(y, ref, what):
    scale = float(np.abs(ref).max())
    err = np.abs(y - ref)
    from tests.conftest import record_measurement
    record_measurement("unet_full:" + what, max_over_scale=float(err.max()) / scale, mean_over_scale=float(err.mean()) / scale)
    assert err.max() < MAX_BAR * scale and err.mean() < MEAN_BAR * scale, (what, float(err.max()), float(err.mean()), scale)
    return float(err.max()) / scale, float(err.mean()) / scale


@pytest.mark.parametrize("tag,B,F,h,w", [("b2f14_72x128", 2, 14, 72, 128), ("b1f25_40x72", 1, 25, 40, 72),
                                         ("b1f25_48x72", 1, 25, 48, 72), ("b2f25_72x128", 2, 25, 72, 128)])
def test_full_width_forward_matches_reference_golden(tag, B, F, h, w, full_unet, gpu, golden_dir):
    path = golden_dir / f"unet_full_{tag}.npz"
    if not path.exists():
        pytest.skip(f"{path.name} not generated (oracle/gen_golden.py unet_full {tag})")
    from syn3r_amd import _lib as L
    L.load().syn3r_gemm_set_tile(0)                       # default dispatch: what bench.py launches
    model, _ = full_unet
    g = np.load(path)
    st = int(g["stride"])
    sample, t, ehs, added = UW.make_inputs(B, F, h, w, seed=F, cross=1024)
    y = model(sample.half().to(gpu), t, ehs.half().to(gpu), added.to(gpu))[0]
    assert y.shape == (B, F, 4, h, w) and y.dtype == torch.float16 and torch.isfinite(y).all()
    yf = y.float().cpu().numpy()
    emax, emean = _compare(yf[..., ::st, ::st], g["out"], tag)
    # whole-output moments (the stored sample is strided): mean |y| and std within the same bar
    assert abs(float(np.abs(yf).mean()) - float(g["mean_abs"])) < 3e-3 * float(np.abs(g["out"]).max())
    assert abs(float(yf.std()) - float(g["std"])) < 1e-2 * float(g["std"])
    print(f"unet_full {tag}: max {emax:.2e} mean {emean:.2e} of scale")


def test_full_width_forward_matches_oracle_small_shape(full_unet, gpu):
    """always-on: the full configuration at [2,2,8,24,32] against the pinned torch-fp32 oracle, run live"""
    from oracle.unet_oracle import UNetOracle
    model, sd = full_unet
    sample, t, ehs, added = UW.make_inputs(2, 2, 24, 32, seed=77, cross=1024)
    ref = UNetOracle(sd, {}).forward(sample, t, ehs, added).numpy()
    y = model(sample.half().to(gpu), t, ehs.half().to(gpu), added.to(gpu))[0].float().cpu().numpy()
    _compare(y, ref, "oracle [2,2,8,24,32]")


def test_full_width_forced_variants_agree_with_default(full_unet, gpu):
    """every contraction-kernel family forced in turn (syn3r_gemm_set_tile) reproduces the default dispatch at a
    mid-size shape to fp16 accuracy: the variants differ only in summation order"""
    from syn3r_amd import _lib as L
    model, _ = full_unet
    lib = L.load()
    sample, t, ehs, added = UW.make_inputs(2, 3, 32, 48, seed=31, cross=1024)
    args = (sample.half().to(gpu), t, ehs.half().to(gpu), added.to(gpu))
    base = model(*args)[0].float()
    scale = float(base.abs().max())
    try:
        for bm in (-128, -256, -320, -322):
            lib.syn3r_gemm_set_tile(bm)
            y = model(*args)[0].float()
            err = (y - base).abs()
            assert float(err.max()) < 2e-2 * scale and float(err.mean()) < 2e-3 * scale, (bm, float(err.max()), scale)
    finally:
        lib.syn3r_gemm_set_tile(0)
