"""Developer tool: per-shape efficiency (TFLOP/s) of the contraction launches of one SVD (step, pass) unit at F frames,
from `tools/unet_breakdown.py <F> detail` output on stdin.   usage: python tools/unet_breakdown.py 25 detail | python tools/shape_eff.py"""
import re
import sys
rows = []
for line in sys.stdin:
    m = re.match(r"(k_gemm\S*?)\[(.*?)\]\s+(\d+) launches\s+([\d.]+) ms", line)
    if not m:
        continue
    kt, n, ms = m.group(1), int(m.group(3)), float(m.group(4))
    f = dict(re.findall(r"([A-Za-z]+)(\d+)", m.group(2)))
    M = int(f["M"])
    if "ffn320" in kt:
        D = int(f["D"]); fl = 2.0 * M * 320 * 3 * D
    elif "lnlin" in kt:
        fl = 2.0 * M * int(f["N"]) * 320
    else:
        fl = 2.0 * M * int(f["N"]) * int(f["K"])
    rows.append((ms, kt, m.group(2), n, fl * n / (ms * 1e-3) / 1e12))
for ms, kt, shp, n, tf in sorted(rows, reverse=True):
    print(f"{kt:22s} {shp:30s} n={n:3d} {ms:8.3f} ms  {tf:7.0f} TFLOP/s  {tf / 2500:.3f}")
