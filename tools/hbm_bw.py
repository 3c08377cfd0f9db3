"""Achievable HBM bandwidth on this box with plain torch kernels (reference points for the HBM-bound kernels)."""
import torch
dev = torch.device("cuda", 0)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (330, 1320):
    x = torch.empty(mb * 1024 * 1024 // 2, dtype=torch.float16, device=dev).normal_()
    y = torch.empty_like(x)
    s = t(lambda: y.copy_(x)); print(f"copy {mb} MB: {2 * x.numel() * 2 / s / 1e12:.2f} TB/s (read+write)")
    s = t(lambda: x.sum()); print(f"read {mb} MB (sum): {x.numel() * 2 / s / 1e12:.2f} TB/s")
    s = t(lambda: y.zero_()); print(f"write {mb} MB (fill): {x.numel() * 2 / s / 1e12:.2f} TB/s")
    s = t(lambda: torch.add(x, y, out=y)); print(f"add {mb} MB (2 reads + 1 write): {3 * x.numel() * 2 / s / 1e12:.2f} TB/s")
