"""Run one UNet contraction shape repeatedly (for rocprofv3 --pmc)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.unet import ops
dev = torch.device("cuda", 0)
H = torch.float16
kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
if kind == "conv":
    x = torch.randn(28, 36, 64, 640, device=dev).to(H)
    w = (torch.randn(640, 3, 3, 640, device=dev) * 0.01).to(H)
    f = lambda: ops.conv3x3(x, w)
elif kind == "lin":
    x = torch.randn(258048, 320, device=dev).to(H)
    w = (torch.randn(2560, 320, device=dev) * 0.05).to(H)
    f = lambda: ops.linear(x, w)
elif kind == "big":      # long-K projection on the wide tile
    x = torch.randn(16128, 5120, device=dev).to(H)
    w = (torch.randn(1280, 5120, device=dev) * 0.02).to(H)
    f = lambda: ops.linear(x, w)
elif kind == "attn":
    qkv = torch.randn(28 * 2304, 1920, device=dev).to(H)
    f = lambda: ops.attention(qkv, 28, 2304, 10)
for _ in range(5):
    f()
torch.cuda.synchronize()
