"""The single-call UNet forward of the C-ABI (syn3r_unet_create / _workspace_bytes / _forward, csrc/unet.hip)
against the Python host graph (syn3r_amd/unet/model.py) on the same checkpoint directory: the same operators in the same order."""
import ctypes as C

import pytest
import torch

from oracle import tiny_checkpoint as TC

pytestmark = pytest.mark.gpu
H = torch.float16


def _abi_forward(lib, handle, sample, t, ehs, added, ctx_group=0):
    from syn3r_amd import _lib
    B, F, _, h, w = sample.shape
    e2 = ehs.reshape(ehs.shape[0], -1).to(H).contiguous()
    shared = B == 1 or ehs.stride(0) == 0
    rows = 1 if shared else B
    need = lib.syn3r_unet_workspace_bytes(handle, B, F, h, w, rows)
    assert need > 0, lib.syn3r_last_error()
    ws = torch.empty(need, dtype=torch.uint8, device=sample.device)
    out = torch.empty(B, F, 4, h, w, dtype=H, device=sample.device)
    ids = added.float().contiguous()
    rc = lib.syn3r_unet_forward(handle, sample.contiguous().data_ptr(), float(t), e2[:rows].contiguous().data_ptr(), rows, ids.data_ptr(),
                                out.data_ptr(), B, F, h, w, ctx_group, ws.data_ptr(), need, _lib.stream_ptr(sample.device))
    _lib.check(rc, "syn3r_unet_forward")
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def tiny(tmp_path_factory, gpu):
    from syn3r_amd import _lib
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    d = TC.write(tmp_path_factory.mktemp("svd")) / "unet"
    model = UNetSpatioTemporalConditionModel.from_pretrained(str(d), gpu, variant="fp16")
    lib = _lib.load()
    handle = C.c_void_p()
    _lib.check(lib.syn3r_unet_create(str(d).encode(), b"fp16", C.byref(handle)), "syn3r_unet_create")
    yield lib, handle, model
    lib.syn3r_unet_destroy(handle)


@pytest.mark.parametrize("B,F,h,w,shared,group", [(2, 5, 16, 24, False, 0), (1, 14, 8, 16, True, 0), (4, 3, 16, 16, False, 2), (2, 25, 8, 8, True, 0)])
def test_unet_abi_forward_equals_python_host(tiny, gpu, B, F, h, w, shared, group):
    lib, handle, model = tiny
    g = torch.Generator().manual_seed(B * 100 + F)
    sample = torch.randn(B, F, 8, h, w, generator=g).to(H).to(gpu)
    ehs = torch.randn(1 if shared else B, 1, 1024, generator=g).to(H).to(gpu)
    if shared and B > 1:
        ehs = ehs.expand(B, -1, -1)
    added = torch.tensor([[6.0, 127.0, 0.02]] * B).to(H).to(gpu)
    t = 1.6377
    ref = model(sample, t, ehs, added, ctx_group=group or None)[0]
    out = _abi_forward(lib, handle, sample, t, ehs, added, ctx_group=group)
    assert out.shape == ref.shape and bool(torch.isfinite(out.float()).all())
    assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()          # same kernels, same order, same bits
    again = _abi_forward(lib, handle, sample, t, ehs, added, ctx_group=group)          # cached position embeddings, reused arena
    assert torch.equal(again, out)


def test_unet_abi_rejects_bad_arguments(tiny, gpu, tmp_path):
    from syn3r_amd import _lib
    lib, handle, _ = tiny
    assert lib.syn3r_unet_workspace_bytes(handle, 1, 40, 8, 8, 1) == 0                # F > 32
    assert lib.syn3r_unet_workspace_bytes(handle, 1, 4, 12, 8, 1) == 0                # h not a multiple of 8
    x = torch.zeros(1, 2, 8, 8, 8, dtype=H, device=gpu)
    e = torch.zeros(1, 1024, dtype=H, device=gpu)
    ids = torch.zeros(1, 3, device=gpu)
    out = torch.empty(1, 2, 4, 8, 8, dtype=H, device=gpu)
    ws = torch.empty(4096, dtype=torch.uint8, device=gpu)
    rc = lib.syn3r_unet_forward(handle, x.data_ptr(), 1.0, e.data_ptr(), 1, ids.data_ptr(), out.data_ptr(), 1, 2, 8, 8, 0, ws.data_ptr(), 4096, None)
    assert rc != 0 and b"workspace" in lib.syn3r_last_error()
    h2 = C.c_void_p()
    assert lib.syn3r_unet_create(str(tmp_path).encode(), None, C.byref(h2)) != 0 and b"config.json" in lib.syn3r_last_error()


def test_unet_abi_full_size_equals_python_host(gpu, tmp_path):
    """The SVD-XT configuration (1.52 B parameters, seeded weights written as a diffusers `unet/` directory) at the bench's
    CFG shape [2, 14, 8, 72, 128]: the C-ABI forward equals the Python host graph bit for bit; the workspace query is what
    the run needs (a smaller workspace is refused, not overrun)."""
    import json
    import time
    from safetensors.torch import save_file
    from syn3r_amd import _lib
    from syn3r_amd.unet.model import UNetSpatioTemporalConditionModel
    model = UNetSpatioTemporalConditionModel().init_random(gpu, seed=3)
    d = tmp_path / "unet"
    d.mkdir()
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in model.config.items()}
    (d / "config.json").write_text(json.dumps(dict(cfg, _class_name="UNetSpatioTemporalConditionModel")))
    save_file({k: v.detach().to("cpu", torch.float16).contiguous() for k, v in model.p.t.items()}, str(d / "diffusion_pytorch_model.fp16.safetensors"))
    lib = _lib.load()
    handle = C.c_void_p()
    t0 = time.time()
    _lib.check(lib.syn3r_unet_create(str(d).encode(), b"fp16", C.byref(handle)), "syn3r_unet_create")
    t_create = time.time() - t0
    try:
        B, F, h, w = 2, 14, 72, 128
        g = torch.Generator().manual_seed(7)
        sample = torch.randn(B, F, 8, h, w, generator=g).to(H).to(gpu)
        ehs = torch.randn(B, 1, 1024, generator=g).to(H).to(gpu)
        added = torch.tensor([[6.0, 127.0, 0.02]] * B).to(H).to(gpu)
        ref = model(sample, 1.6377, ehs, added)[0]
        out = _abi_forward(lib, handle, sample, 1.6377, ehs, added)
        assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()
        need = lib.syn3r_unet_workspace_bytes(handle, B, F, h, w, B)
        ws = torch.empty(need, dtype=torch.uint8, device=gpu)
        o2 = torch.empty_like(out)
        ids = added.float().contiguous()
        e2 = ehs.reshape(B, -1).contiguous()
        args = (handle, sample.data_ptr(), 1.6377, e2.data_ptr(), B, ids.data_ptr(), o2.data_ptr(), B, F, h, w, 0)
        rc = lib.syn3r_unet_forward(*args, ws.data_ptr(), need // 2, _lib.stream_ptr(gpu))
        assert rc != 0 and b"workspace" in lib.syn3r_last_error()
        # timing: the C-ABI call against the Python host (eager), same kernels
        def timed(f, n=3):
            f(); torch.cuda.synchronize()
            t = time.time()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            return (time.time() - t) / n * 1e3
        ms_abi = timed(lambda: _lib.check(lib.syn3r_unet_forward(*args, ws.data_ptr(), need, _lib.stream_ptr(gpu)), "forward"))
        ms_py = timed(lambda: model(sample, 1.6377, ehs, added))
        print(f"\n[unet abi] create {t_create:.1f} s, workspace {need / 2**30:.2f} GiB, forward {ms_abi:.1f} ms (C-ABI) vs {ms_py:.1f} ms (Python host, eager)")
        assert torch.equal(o2, ref)
    finally:
        lib.syn3r_unet_destroy(handle)
