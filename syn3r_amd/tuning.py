"""Developer switches for A/B measurements of the host graph: plain module attributes.

The product never reads the environment to decide which kernels run (a stray variable in a
maintainer's shell must not change the launch sequence): a measuring script under `tools/` sets
`syn3r_amd.tuning.FLAGS[...]` itself, or calls `from_env()` to take them from `SYN3R_*` variables
explicitly.  The library's own dispatch switches exist only in `-DSYN3R_TUNING` builds (csrc/common.h).
"""
from __future__ import annotations

import os

FLAGS = {
    "ff_fused": True,       # FeedForward at C = 320 in one kernel (k_ffn320r)
    "ff_ln": True,          # norm3 / norm_in inside the fused feed-forward
    "ff_tiled": True,       # gated intermediate of the two-kernel feed-forward in the A-tiled workspace
    "ln_qkv": True,         # norm1 + stacked q / k / v projection in one kernel (k_lnlin320)
    "splitk": True,         # split-K for the level-3 contractions
    "unet_cat": False,      # materialise torch.cat([hidden, skip]) instead of the two-source operands
    "unet_graph": False,    # replay captured UNet launch sequences (hipGraph)
    "two_streams": True,    # independent launch sequences of a two-pass step on their own HIP streams
    "merge_passes": True,   # both passes of a step in one stack of UNet launches
    "ff_g256": True,        # net.0 of the two-kernel feed-forward on the 256 x 256 tile (k_gemm_g256) where whole tiles fit
    "gn_epilogue": True,    # GroupNorm statistics from the producing contraction's epilogue (no k_gn_stats pass)
}

_ENV = {"ff_fused": "SYN3R_FF_FUSED", "ff_ln": "SYN3R_FF_LN", "ff_tiled": "SYN3R_FF_TILED", "ln_qkv": "SYN3R_LN_QKV",
        "splitk": "SYN3R_SPLITK", "unet_cat": "SYN3R_UNET_CAT", "unet_graph": "SYN3R_UNET_GRAPH",
        "two_streams": "SYN3R_TWO_STREAMS", "merge_passes": "SYN3R_MERGE_PASSES", "gn_epilogue": "SYN3R_GN_EPILOGUE", "ff_g256": "SYN3R_FF_G256"}


def from_env() -> dict:
    """tools/ only: take the switches from their SYN3R_* environment variables ("0" / "1")."""
    for key, name in _ENV.items():
        v = os.environ.get(name)
        if v is not None:
            FLAGS[key] = v != "0"
    return FLAGS
