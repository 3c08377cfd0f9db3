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
    assert rc != 0 and ("outside the file" in err or "exceeds the file" in err)   # the entry points past the end of the file
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


def test_unet_create_survives_hostile_safetensors_headers(lib, tmp_path):
    """Dimensions, offsets and escapes of the header are file contents: negative / overflowing / fractional / huge values, a
    truncated \\u escape and a number at the very end of the header must end in an error return (never an exception through
    extern "C", an allocation sized by the header, or a read past the mapping)."""
    cfg = dict(in_channels=8, out_channels=4, block_out_channels=[64, 128], num_attention_heads=[1, 2], layers_per_block=1,
               down_block_types=["CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"],
               up_block_types=["UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"])
    (tmp_path / "config.json").write_text(json.dumps(cfg))
    st = tmp_path / "diffusion_pytorch_model.safetensors"
    bad_entries = [
        {"dtype": "F16", "shape": [-1, 8, 3, 3], "data_offsets": [0, 16]},                       # negative dimension
        {"dtype": "F16", "shape": [1 << 40, 1 << 40], "data_offsets": [0, 16]},                  # product overflows 2^64
        {"dtype": "F16", "shape": [1 << 33, 1 << 33], "data_offsets": [0, 16]},                  # 2^66 elements
        {"dtype": "F16", "shape": [2.5, 8], "data_offsets": [0, 16]},                            # fractional
        {"dtype": "F16", "shape": [64, 8, 3, 3], "data_offsets": [-16, 16]},                     # negative offset
        {"dtype": "F16", "shape": [64, 8, 3, 3], "data_offsets": [0, 1e30]},                     # far beyond 2^53
        {"dtype": "F16", "shape": [1 << 30], "data_offsets": [0, 16]},                           # count x 2 != the 16 bytes it names
        {"dtype": "F16", "shape": [8] * 9, "data_offsets": [0, 16]},                             # too many dimensions
        {"dtype": "F16", "shape": "oops", "data_offsets": [0, 16]},
    ]
    for ent in bad_entries:
        _safetensors(st, {"conv_in.weight": ent}, b"\0" * 16)
        rc, err, h = _create(lib, tmp_path, variant=None)
        assert rc != 0 and not h.value, ent
        assert any(k in err for k in ("shape", "malformed", "outside the file", "size mismatch", "exceeds", "JSON", "missing tensor")), (ent, err)
    # raw headers json.dumps would not write: a truncated \u escape at the end, and a number that runs to the last header byte
    for raw in (b'{"a\\u12', b'{"conv_in.weight":{"dtype":"F16","shape":[1],"data_offsets":[0,2', b'{"x":' + b"9" * 400):
        st.write_bytes(struct.pack("<Q", len(raw)) + raw)
        rc, err, h = _create(lib, tmp_path, variant=None)
        assert rc != 0 and not h.value and "JSON" in err, (raw, err)
    # a header length that covers the whole file (no payload) with a sane entry: outside the file, not a wrapped offset
    _safetensors(st, {"conv_in.weight": {"dtype": "F16", "shape": [4], "data_offsets": [(1 << 52) - 8, (1 << 52)]}})
    rc, err, h = _create(lib, tmp_path, variant=None)
    assert rc != 0 and "outside the file" in err
