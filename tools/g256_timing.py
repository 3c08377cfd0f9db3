"""Where a k_gemm_g256 wavefront spends a tile (developer tool).  Needs a timing build:
    tools/build_variant.sh g256time -DSYN3R_TIMING ; SYN3R_LIB_OVERRIDE=abtmp/libg256time.so python tools/g256_timing.py
Prints s_memtime ticks per tile for wavefronts 0 and 4 of one block: k-tiles 0-1 (no counted wait), the counted wait of k-tile 2 (the
first that includes the previous tile's store acknowledgements), the counted waits of the later k-tiles, the barriers, the rest of the
k-loop, boundary + gate + the wait in front of the stores, the store issue."""
import ctypes
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401
from syn3r_amd import _lib
from syn3r_amd.unet import ops

dev = torch.device("cuda", 0)
H = torch.float16
lib = ctypes.CDLL(str(_lib._LIB_PATH))
if not hasattr(lib, "syn3r_debug_g256_timing"):
    sys.exit("library built without -DSYN3R_TIMING")
NAMES = ["kt0-1", "wait@kt2", "wait@kt>2", "barriers", "k-loop rest", "boundary+gate+wait", "store issue"]
g = torch.Generator().manual_seed(1)
for M, C, D in ((64512, 640, 2560), (16128, 1280, 5120)):
    x = torch.randn(M, C, generator=g).to(H).to(dev)
    w1 = (torch.randn(2 * D, C, generator=g) * C ** -0.5).to(H).to(dev)
    b1 = torch.randn(2 * D, generator=g).to(H).to(dev)
    w2 = (torch.randn(C, D, generator=g) * D ** -0.5).to(H).to(dev)
    wp, bp, _ = ops.pack_geglu(w1, b1)
    w64, b64, _ = ops.pack_geglu64(w1, b1)
    for _ in range(5):
        ops.feedforward(x, wp, bp, D, w2, None, packed64=(w64, b64))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    lib.syn3r_debug_g256_timing(out)          # (the last timed launch: net.0 ran before net.2, which does not touch the buffer)
    print(f"[{M},{2 * D},{C}]  ticks per tile")
    for wv in (0, 1, 4, 5):
        nt = max(out[wv * 8 + 7], 1)
        v = [out[wv * 8 + i] / nt for i in range(7)]
        print(f"  wave {wv} ({nt} tiles): " + "  ".join(f"{n}={t:7.0f}" for n, t in zip(NAMES, v)) + f"  total={sum(v):8.0f}")
