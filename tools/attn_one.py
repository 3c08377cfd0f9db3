"""A few spatial-attention launches at the UNet's level-0 shape (developer tool: target of rocprofv3 --pmc)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.unet import ops
dev = torch.device("cuda", 0)
BF, hw, heads = (int(sys.argv[1]) if len(sys.argv) > 1 else 28), 72 * 128, 5
qkv = torch.randn(BF * hw, 3 * 64 * heads, device=dev).to(torch.float16)
for _ in range(2):
    o = ops.attention(qkv, BF, hw, heads)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); o = ops.attention(qkv, BF, hw, heads); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
print(f"attn L0 BF={BF}: {ms:.3f} ms  {4.0 * BF * heads * hw * hw * 64 / ms / 1e9:.1f} TFLOP/s")
