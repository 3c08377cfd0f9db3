"""Which LPIPS layer carries the fp16 error (VERDICT r03, weak points): value and image-gradient of the HIP path against the
float64 oracle with ONE layer's linear weights active at a time (the others zero), 136 x 240 test images, seeded weights.
python tools/lpips_layers.py"""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import lpips_oracle as LO
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.gs.lpips import LPIPS

dev = torch.device("cuda", 0)
H, W = 136, 240
g = torch.Generator().manual_seed(H)
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
base = torch.stack([0.5 + 0.4 * torch.sin(xs / 7 + c) * torch.cos(ys / 5 - c) for c in range(3)])
a = (base + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
b = (base.roll(2, 2) + 0.08 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
shapes = LPIPS().parameter_shapes()
gw = torch.Generator().manual_seed(3)
sd = {}
for k, shape in shapes.items():
    if k.startswith("lin"):
        sd[k] = torch.rand(shape, generator=gw) * 0.2 + 0.01
    elif k.endswith(".bias"):
        sd[k] = 0.05 * torch.randn(shape, generator=gw)
    else:
        sd[k] = torch.randn(shape, generator=gw) * math.sqrt(2.0 / (shape[1] * 9))
names = ["relu1_2 (64 ch, full res)", "relu2_2 (128 ch, 1/2)", "relu3_3 (256 ch, 1/4)", "relu4_3 (512 ch, 1/8)", "relu5_3 (512 ch, 1/16)"]
print("layer                         value: HIP / oracle   rel err |  gradient: rel err   cosine | share of the full loss")
full = float(LO.lpips(a.double(), b.double(), sd))
for only in list(range(5)) + [None]:
    s2 = {k: (v if (not k.startswith("lin") or only is None or k.startswith(f"lin{only}.")) else torch.zeros_like(v)) for k, v in sd.items()}
    m = LPIPS().load_state_dict({k: v.clone() for k, v in s2.items()}, dev)
    pred = a.to(dev).requires_grad_(True)
    loss = m(pred, b.to(dev))
    loss.backward()
    ao = a.double().requires_grad_(True)
    ref = LO.lpips(ao, b.double(), s2)
    ref.backward()
    gh, go = pred.grad.double().cpu(), ao.grad
    rel = float((gh - go).norm() / go.norm())
    cos = float((gh * go).sum() / (gh.norm() * go.norm()))
    name = names[only] if only is not None else "all five layers"
    print(f"{name:28s} {float(loss):.6f} / {float(ref):.6f}  {abs(float(loss) - float(ref)) / abs(float(ref)):8.2e} | {rel:18.2e} {cos:8.5f} | {float(ref) / full:6.3f}")
