"""UNet host module: parameter table vs the reference (CPU) and forward vs the reference's golden output (GPU)."""
import ast

import numpy as np
import pytest
import torch

from oracle import unet_weights as UW


def test_parameter_names_and_shapes_match_reference(golden_dir):
    """The HIP UNet declares exactly the reference module's state_dict (names and shapes), so a diffusers
    checkpoint loads unchanged."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    g = np.load(golden_dir / "unet_small.npz")
    ref = {str(n): ast.literal_eval(str(s)) for n, s in zip(g["names"], g["shapes"])}
    mine = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG).parameter_shapes()
    assert set(mine) == set(ref)
    for k in ref:
        assert tuple(mine[k]) == tuple(ref[k]), k


def test_default_config_is_svd_xt_sized():
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    m = UNetSpatioTemporalConditionModel()
    assert m.num_parameters() == 1_524_623_082      # SURVEY.md §2.2 / BASELINE.md §2
    with pytest.raises(ValueError):
        UNetSpatioTemporalConditionModel(down_block_types=("DownBlockSpatioTemporal",) * 3)
    with pytest.raises(Exception):
        m.forward(torch.zeros(1, 2, 8, 8, 8), 1.0, torch.zeros(1, 1, 1024), torch.zeros(1, 3))   # weights not loaded


@pytest.mark.gpu
@pytest.mark.parametrize("tag,B,F,h,w", [("b2f5", 2, 5, 16, 24), ("b1f14", 1, 14, 8, 16)])
def test_forward_matches_reference_golden(tag, B, F, h, w, gpu, golden_dir):
    """fp16 HIP forward vs the reference's fp32 CPU forward on identical (fp16-representable) weights.
    Tolerance: fp16 activations through ~60 layers -> 2e-2 of the output scale."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    g = np.load(golden_dir / "unet_small.npz")
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    model.load_state_dict(UW.make_state_dict(model.parameter_shapes()), gpu)
    sample, t, ehs, added = UW.make_inputs(B, F, h, w, seed=F)
    y = model(sample.half().to(gpu), t, ehs.half().to(gpu), added.to(gpu))[0]
    assert y.shape == (B, F, 4, h, w) and y.dtype == torch.float16
    ref = torch.from_numpy(g[f"{tag}_out"])
    err = (y.float().cpu() - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() < 3e-2 * scale, (err.max().item(), scale)
    assert err.mean().item() < 3e-3 * scale, (err.mean().item(), scale)


def test_unet_oracle_matches_reference_golden(golden_dir):
    """oracle/unet_oracle.py (torch fp32 CPU restatement) is pinned by the reference module's own outputs."""
    from oracle.unet_oracle import UNetOracle
    g = np.load(golden_dir / "unet_small.npz")
    shapes = {str(n): ast.literal_eval(str(s)) for n, s in zip(g["names"], g["shapes"])}
    orc = UNetOracle(UW.make_state_dict(shapes), UW.SMALL_CONFIG)
    for tag, (B, F, h, w) in {"b2f5": (2, 5, 16, 24), "b1f14": (1, 14, 8, 16)}.items():
        y = orc.forward(*UW.make_inputs(B, F, h, w, seed=F))
        torch.testing.assert_close(y, torch.from_numpy(g[f"{tag}_out"]), atol=2e-4, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,h,w", [(2, 7, 16, 40), (1, 25, 8, 8), (2, 3, 48, 16)])
def test_forward_matches_oracle_on_other_shapes(B, F, h, w, gpu):
    """Shapes the golden file does not hold (the reference's 25 frames, non-square and odd frame counts), against
    the pinned CPU oracle; same tolerance as the golden comparison."""
    from oracle.unet_oracle import UNetOracle
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    sd = UW.make_state_dict(model.parameter_shapes())
    model.load_state_dict(sd, gpu)
    sample, t, ehs, added = UW.make_inputs(B, F, h, w, seed=100 + F)
    ref = UNetOracle(sd, UW.SMALL_CONFIG).forward(sample, t, ehs, added)
    y = model.forward(sample.to(gpu).half(), t, ehs.to(gpu).half(), added.to(gpu))[0].float().cpu()
    scale = float(ref.abs().max())
    err = (y - ref).abs()
    assert float(err.max()) < 3e-2 * scale and float(err.mean()) < 3e-3 * scale, (float(err.max()), float(err.mean()), scale)


@pytest.mark.gpu
def test_forward_is_deterministic_and_context_interleave(gpu):
    """Same input -> bit-identical output.  Batch items only interact through the reference's
    pixel-major/batch-minor temporal context layout (transformer_temporal.py:310-317): with equal
    contexts the items are independent, with different contexts they are not."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    model.load_state_dict(UW.make_state_dict(model.parameter_shapes()), gpu)
    sample, t, ehs, added = UW.make_inputs(2, 5, 16, 24, seed=3)
    s, e, a = sample.half().to(gpu), ehs.half().to(gpu), added.to(gpu)
    y1 = model(s, t, e, a)[0]
    y2 = model(s, t, e, a)[0]
    assert torch.equal(y1, y2)
    y_single = model(s[1:], t, e[1:], a[1:])[0]
    assert not torch.allclose(y_single.float(), y1[1:].float(), atol=2e-3, rtol=1e-2)
    e_same = e[1:].repeat(2, 1, 1)
    y_same = model(s, t, e_same, a)[0]
    assert torch.allclose(y_single.float(), y_same[1:].float(), atol=2e-3, rtol=1e-2)


@pytest.mark.gpu
def test_shared_context_batch_equals_single_items(gpu):
    """The guidance tiles of the Post pipeline run as batch-of-2 forwards with ONE (stride-0 expanded) context:
    every item must come out as its own B = 1 forward does — also where h*w is odd at the coarsest level (5x9),
    which the batch-interleaved context path cannot serve."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    model.load_state_dict(UW.make_state_dict(model.parameter_shapes()), gpu)
    sample, t, ehs, added = UW.make_inputs(2, 5, 40, 72, seed=9)
    x = sample.to(gpu).half()
    e1 = ehs[:1].to(gpu).half()
    a1 = added[:1].to(gpu)
    both = model(x, t, e1.expand(2, -1, -1), a1.expand(2, -1).contiguous())[0]
    for n in range(2):
        one = model(x[n:n + 1].contiguous(), t, e1, a1)[0]
        err = (both[n:n + 1].float() - one.float()).abs()
        scale = float(one.float().abs().max())
        assert float(err.max()) < 2e-2 * scale and float(err.mean()) < 2e-3 * scale, (float(err.max()), scale)
    with pytest.raises(NotImplementedError):        # two DIFFERENT contexts need the interleave, which needs even h*w
        model(x, t, ehs.to(gpu).half(), added.to(gpu))


@pytest.mark.gpu
def test_ctx_group_stack_equals_separate_calls(gpu):
    """forward(..., ctx_group=G): a batch that stacks independent batch-of-G calls (the two passes of a denoising step in
    one launch sequence) returns what the separate calls return — including the reference's batch-interleaved temporal
    context, which couples the samples INSIDE a call and must not couple the stacked calls."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    model.load_state_dict(UW.make_state_dict(model.parameter_shapes()), gpu)
    sample, t, ehs, added = UW.make_inputs(4, 5, 16, 32, seed=21)       # 2 x 4 pixels at the coarsest level: divisible by 4
    s, e, a = sample.half().to(gpu), ehs.half().to(gpu), added[:1].repeat(4, 1).to(gpu)
    close = lambda x, y: torch.allclose(x.float(), y.float(), atol=2e-3 * float(y.float().abs().max()), rtol=1e-2)
    both = model(s, t, e, a, ctx_group=2)[0]
    for k in range(2):
        sep = model(s[2 * k:2 * k + 2].contiguous(), t, e[2 * k:2 * k + 2].contiguous(), a[:2])[0]
        assert close(both[2 * k:2 * k + 2], sep)
    whole = model(s, t, e, a)[0]                              # one batch-of-4 call interleaves over all four contexts: different
    assert not close(whole, both)
    ones = model(s[:2].contiguous(), t, e[:2].contiguous(), a[:2], ctx_group=1)[0]        # G = 1: every sample its own B = 1 call
    for k in range(2):
        assert close(ones[k:k + 1], model(s[k:k + 1].contiguous(), t, e[k:k + 1].contiguous(), a[:1])[0])
    with pytest.raises(ValueError):
        model(s, t, e, a, ctx_group=3)


@pytest.mark.gpu
def test_captured_forward_replays_bit_identical(gpu):
    """`forward_graphed`: the launch sequence of one (shape, context) captured into a hipGraph and replayed gives the eager
    forward bit for bit, also after the inputs and the timestep change; a new shape gets its own graph."""
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel(**UW.SMALL_CONFIG)
    model.load_state_dict(UW.make_state_dict(model.parameter_shapes()), gpu)
    sample, t, ehs, added = UW.make_inputs(2, 5, 16, 24, seed=8)
    s, e, a = sample.half().to(gpu), ehs.half().to(gpu), added.to(gpu)
    for tt, scale in ((t, 1.0), (0.37, 0.5), (torch.tensor(1.2, device=gpu), 2.0)):
        x = (s * scale).contiguous()
        assert torch.equal(model.forward_graphed(x, tt, e, a)[0], model(x, tt, e, a)[0])
    assert len(model._graphs) == 2                     # one per timestep dtype (float64 scalars, the float32 tensor)
    # the graph entry pins what its launches address outside the graph's pool: drop the model's context store and the
    # library's scratch cache, churn the allocator, replay - the result is still the eager one
    from syn3r_amd import _lib as L
    ref = model(s, t, e, a)[0]
    model._ctx_store.clear()
    L._ws_cache.clear()
    junk = [torch.full((1 << 22,), 7.0, device=gpu) for _ in range(8)]
    assert torch.equal(model.forward_graphed(s, t, e, a)[0], ref)
    del junk
    model.invalidate_context_cache()
    assert len(model._graphs) == 0                     # a graph does not outlive the contexts it was captured with
    assert torch.equal(model.forward_graphed(s, t, e, a)[0], ref) and len(model._graphs) == 1
    x2 = s[..., :16].contiguous()                      # 16 x 16: an even pixel count at the coarsest level too
    assert torch.equal(model.forward_graphed(x2, t, e, a)[0], model(x2, t, e, a)[0]) and len(model._graphs) == 2
    s4 = torch.cat([s, s * 0.7])
    e4 = torch.cat([e, e.flip(0)]).contiguous()
    a4 = torch.cat([a, a])
    assert torch.equal(model.forward_graphed(s4, t, e4, a4, ctx_group=2)[0], model(s4, t, e4, a4, ctx_group=2)[0])
