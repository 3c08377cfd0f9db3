"""CPU tier of the single-call UNet ABI (csrc/unet.hip): the checkpoint reader's error paths - every one of them ends before
the first device call, so no GPU is needed."""
import ctypes as C
import json
import struct

import pytest


@pytest.fixture(scope="module")
def lib():
    from syn3r_amd import _lib
    return _lib.load()


def _create(lib, d, variant=b"fp16"):
    h = C.c_void_p()
    rc = lib.syn3r_unet_create(str(d).encode(), variant, C.byref(h))
    return rc, lib.syn3r_last_error().decode(), h


def _safetensors(path, header: dict, payload: bytes = b""):
    hb = json.dumps(header).encode()
    path.write_bytes(struct.pack("<Q", len(hb)) + hb + payload)


def test_unet_create_reports_what_is_wrong_with_a_directory(lib, tmp_path):
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "config.json" in err
    (tmp_path / "config.json").write_text("{ not json")
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "JSON" in err
    (tmp_path / "config.json").write_text(json.dumps(dict(block_out_channels=[64, 100], num_attention_heads=[1, 2])))
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "head dim" in err                                   # 100 channels is not 64 x heads
    cfg = dict(in_channels=8, out_channels=4, block_out_channels=[64, 128], num_attention_heads=[1, 2], layers_per_block=1,
               down_block_types=["CrossAttnDownBlockSpatioTemporal", "NoSuchBlock"],
               up_block_types=["UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"])
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "NoSuchBlock" in err
    cfg["down_block_types"][1] = "DownBlockSpatioTemporal"
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "safetensors" in err                                # no weights file at all
    (tmp_path / "diffusion_pytorch_model.fp16.safetensors").write_bytes(b"\x01\x02\x03")
    rc, err, _ = _create(lib, tmp_path)
    assert rc != 0 and "not a safetensors file" in err                     # too short to hold a header
    (tmp_path / "diffusion_pytorch_model.fp16.safetensors").unlink()
    _safetensors(tmp_path / "diffusion_pytorch_model.safetensors", {"conv_in.weight": {"dtype": "F16", "shape": [64, 8, 3, 3], "data_offsets": [0, 9216]}})
    rc, err, _ = _create(lib, tmp_path, variant=None)
    assert rc != 0 and "outside the file" in err                           # the entry points past the end of the file
    _safetensors(tmp_path / "diffusion_pytorch_model.safetensors",
                 {"conv_in.weight": {"dtype": "I8", "shape": [4], "data_offsets": [0, 4]}, "__metadata__": {"format": "pt"}}, b"\0" * 4)
    rc, err, _ = _create(lib, tmp_path, variant=None)
    assert rc != 0 and "unsupported dtype" in err
    _safetensors(tmp_path / "diffusion_pytorch_model.safetensors",
                 {"conv_in.bias": {"dtype": "F32", "shape": [2], "data_offsets": [0, 8]}}, b"\0" * 8)
    rc, err, _ = _create(lib, tmp_path, variant=None)
    assert rc != 0 and "missing tensor" in err                             # a readable file that is not this model's


def test_unet_handles_are_checked(lib):
    junk = (C.c_char * 4096)()
    assert lib.syn3r_unet_workspace_bytes(C.cast(junk, C.c_void_p), 1, 2, 8, 8, 1) == 0
    assert lib.syn3r_unet_destroy(C.cast(junk, C.c_void_p)) != 0 and b"live handle" in lib.syn3r_last_error()
    assert lib.syn3r_unet_destroy(None) == 0
