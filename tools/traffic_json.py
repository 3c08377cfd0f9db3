"""Build profiles/rNN/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench.

usage: python tools/traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [<pmc_families.json>] [<unet_shapes.txt>]
(the optional fourth file, written by tools/pmc_families.py from the SQ / GRBM passes of the same build, adds `mfma_busy`;
the optional fifth, the per-shape table of tools/unet_breakdown.py, adds the ALGORITHMIC bytes per launch of every
contraction kernel - A once, W once, out once, residual once - and with them `traffic_ratio` = counted / algorithmic)

Corrections follow MI355X_MICROARCH.md (HBM): counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of 16 B/lane streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming
stores and float atomics.  Values are averaged over all launches of a kernel (type) in the profiled run.
The contraction FAMILY `k_gemm` = every tile kernel of csrc/gemm.hip including the fused feed-forward k_ffn320 and
excluding k_gemm_skinny (M <= 16 weight streams: ~100 near-empty launches that would dilute the per-launch average).
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import _devlib  # noqa: F401  (SYN3R_LIB_OVERRIDE=<other build>: explicit, tool-side)
from syn3r_amd.pipeline.svd_step import source_id  # noqa: E402

OTHER = {"k_attn_spatial": "k_attn_spatial", "k_render(": "k_render", "k_render_bwd": "k_render_bwd",
         "k_scatter": "k_scatter", "k_preprocess_bwd": "k_preprocess_bwd"}


def ktype(name):
    """rocprofv3 kernel name -> the tracer's kernel type (bench.py / unet_breakdown.py spell them the same way)."""
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(" ", "")
    if "k_ffn320" in n:
        return "k_gemm_ffn320"
    if "k_lnlin320" in n:
        return "k_gemm_lnlin320"
    m = re.search(r"k_gemm\w*(<[^>]*>)?", n)
    if m:
        t = m.group(0)
        if t.startswith("k_gemm_z"):                      # k_gemm_z<TWOSRC, MODE[, ...]> -> the tracer's k_gemm_z<MODE>
            a = re.findall(r"-?\d+|true|false", t[len("k_gemm_z"):])
            return f"k_gemm_z<{a[1] if len(a) > 1 else 0}>"
        return "k_gemm_skinny" if t.startswith("k_gemm_skinny") else t
    for pat, fam in OTHER.items():
        if pat in name:
            return fam
    return None


def collect(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        t = ktype(r["Kernel_Name"])
        if t is not None:
            tot[t][0] += 1
            tot[t][1] += float(r["Counter_Value"])
    return tot


def algorithmic_bytes(shapes_txt):
    """{kernel type: launch-weighted mean algorithmic bytes per launch} from `unet_breakdown.py <F> detail` lines such as
    `k_gemm_dma<1,256>[M258048,N320,K2880,e1]   5 launches ...` (e1: + a residual read, e2: GEGLU, half the columns written)."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    for line in open(shapes_txt):
        m = re.match(r"(k_gemm\S*?)\[(.*?)\]\s+(\d+) launches", line)
        if not m:
            continue
        kt, n = m.group(1), int(m.group(3))
        f = dict(re.findall(r"([A-Za-z]+)(\d+)", m.group(2)))
        M, e = int(f["M"]), int(f.get("e", 0))
        if kt == "k_gemm_ffn320":
            D = int(f["D"])
            b = 2 * (M * 320 * 2 + 3 * D * 320 + (M * 320 if e else 0))
        elif kt == "k_gemm_lnlin320":                      # LayerNorm + projection at K = 320: x once, W once, out once
            N = int(f["N"])
            b = 2 * (M * 320 + N * 320 + M * N)
        else:
            N, K = int(f["N"]), int(f["K"])
            taps = 9 if "<1" in kt else (3 if "<2" in kt else 1)            # implicit GEMM (MODE 1 / 2): every input element counted once
            b = 2 * (M * (K // taps) + N * K + M * (N // 2 if e == 2 else N) + (M * N if e == 1 else 0))
        acc[kt][0] += n
        acc[kt][1] += float(b) * n
    return {k: v[1] / v[0] for k, v in acc.items() if v[0]}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
               "`python3 bench.py --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace`; counters are KB; FETCH_SIZE "
               "doubled (gfx950 tallies the 128-B requests of 16 B/lane streams at 64 B, MI355X_MICROARCH.md HBM); "
               "WRITE_SIZE as read; per-launch averages over every launch of the kernel (type); the family k_gemm = the tile "
               "kernels + k_ffn320, without k_gemm_skinny",
       "_source_id": source_id()}      # the kernel sources this pass profiled: bench.py reports the traffic only for that build
by = {}
for t in sorted(set(fetch) | set(write)):
    n = max(fetch[t][0], write[t][0], 1)
    f_kb, w_kb = fetch[t][1] / max(fetch[t][0], 1), write[t][1] / max(write[t][0], 1)
    by[t] = {"launches": n, "fetch_kb_raw": round(f_kb, 1), "write_kb": round(w_kb, 1), "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024)}
fam = [t for t in by if t.startswith("k_gemm") and t != "k_gemm_skinny"]
for t in by:
    if not t.startswith("k_gemm"):
        out[t] = by[t]
if fam:
    n = sum(by[t]["launches"] for t in fam)
    out["k_gemm"] = {"launches": n, "members": fam,
                     "fetch_kb_raw": round(sum(by[t]["fetch_kb_raw"] * by[t]["launches"] for t in fam) / n, 1),
                     "write_kb": round(sum(by[t]["write_kb"] * by[t]["launches"] for t in fam) / n, 1),
                     "hbm_bytes_per_launch": int(sum(by[t]["hbm_bytes_per_launch"] * by[t]["launches"] for t in fam) / n)}
out["by_kernel"] = {t: by[t] for t in by if t.startswith("k_gemm")}
shapes = next((a for a in sys.argv[4:] if a.endswith(".txt")), None)
if shapes and fam:
    alg = algorithmic_bytes(shapes)
    num = den = 0.0
    worst = None
    for t in fam:
        if t in alg:
            r = by[t]["hbm_bytes_per_launch"] / alg[t]
            out["by_kernel"][t].update(algorithmic_bytes_per_launch=int(alg[t]), traffic_ratio=round(r, 3))
            num += by[t]["hbm_bytes_per_launch"] * by[t]["launches"]
            den += alg[t] * by[t]["launches"]
            if worst is None or r > worst[1]:
                worst = (t, r)
    if den > 0:
        out["k_gemm"].update(algorithmic_bytes_per_launch=int(den / sum(by[t]["launches"] for t in fam if t in alg)),
                             traffic_ratio={"family": round(num / den, 3), "worst_kernel": worst[0], "worst": round(worst[1], 3)})
famjson = next((a for a in sys.argv[4:] if a.endswith(".json")), None)
if famjson:
    pf = json.load(open(famjson))
    gemm = [v for k, v in pf.items() if (k.startswith("k_gemm") and k != "k_gemm_skinny") or k in ("k_ffn320", "k_lnlin320")]
    # MFMA busy of the whole contraction family: launch-weighted by the chip-active cycles of each member
    num = sum(v["per_launch"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * v["launches"] for v in gemm)
    den = sum(v["per_launch"].get("GRBM_GUI_ACTIVE", 0.0) * v["launches"] for v in gemm) / 8.0 * 1024.0
    if den > 0 and "k_gemm" in out:
        out["k_gemm"]["mfma_busy"] = round(num / den, 4)
    for name in ("k_attn_spatial", "k_render_bwd", "k_render"):
        if name in out and name in pf and "mfma_busy" in pf[name]:
            out[name]["mfma_busy"] = pf[name]["mfma_busy"]
        # executed vector wave-instructions per launch: bench.py's VALU-issue roof of the blend kernels
        if name in pf and "SQ_INSTS_VALU" in pf[name].get("per_launch", {}):
            out.setdefault(name, {})["valu_insts_per_launch"] = pf[name]["per_launch"]["SQ_INSTS_VALU"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
