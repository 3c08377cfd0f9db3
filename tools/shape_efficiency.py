"""Per-shape matrix efficiency of the contraction kernels of one SVD unit, from the detail table tools/unet_breakdown.py prints
(`python tools/unet_breakdown.py 14 detail > unet_shapes.txt`): FLOPs of a launch from its name (2 M N K; the fused feed-forward
2 M (2 D 320 + 320 D); LayerNorm + projection 2 M N 320; gated projections count the packed N, hidden and gate), TFLOP/s over the
shape's summed device time and the fraction of the 2.5 PFLOP/s dense fp16 peak.  Part of tools/collect_profiles.sh, so the table
always belongs to the build the round's bench line was measured on.
usage: python tools/shape_efficiency.py unet_shapes.txt > shape_efficiency.txt"""
import re
import sys

PEAK = 2500.0
rows, tot_ms, tot_fl = [], 0.0, 0.0
for line in open(sys.argv[1]):
    m = re.match(r"(k_gemm_[\w<>,/]+)\[([^\]]+)\]\s+(\d+) launches\s+([\d.]+) ms", line)
    if not m:
        continue
    name, shape, n, ms = m.group(1), m.group(2), int(m.group(3)), float(m.group(4))
    d = {k: int(v) for k, v in re.findall(r"([MNKD])(\d+)", shape)}
    if "ffn320" in name:
        fl = 2.0 * d["M"] * (2 * d["D"] * 320 + 320 * d["D"])
    elif "lnlin320" in name:
        fl = 2.0 * d["M"] * d["N"] * 320
    elif {"M", "N", "K"} <= set(d):
        fl = 2.0 * d["M"] * d["N"] * d["K"]
    else:
        continue
    fl *= n
    rows.append((ms, name, shape, n, fl))
    tot_ms += ms
    tot_fl += fl
for ms, name, shape, n, fl in sorted(rows, reverse=True):
    tf = fl / ms / 1e9
    print(f"{name:24s} {shape:34s} n={n:3d} {ms:8.3f} ms {tf:8.0f} TFLOP/s  {tf / PEAK:.3f}")
if tot_ms:
    print(f"{'contraction family':24s} {'(every launch listed above)':34s}       {tot_ms:8.3f} ms {tot_fl / tot_ms / 1e9:8.0f} TFLOP/s  {tot_fl / tot_ms / 1e9 / PEAK:.3f}")
