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
