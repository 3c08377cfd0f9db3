"""Build profiles/rNN/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench.

usage: python tools/traffic_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [<pmc_families.json>]
(the optional fourth file, written by tools/pmc_families.py from the SQ / GRBM passes of the same build, adds `mfma_busy`)

Corrections follow MI355X_MICROARCH.md (HBM): counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of 16 B/lane streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16 B/lane streaming
stores and float atomics.  Values are averaged over all launches of a kernel family in the profiled run.
"""
import collections
import csv
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from syn3r_amd.pipeline.svd_step import source_id  # noqa: E402

FAMILIES = {"k_gemm": "k_gemm", "k_attn_spatial": "k_attn_spatial", "k_render(": "k_render", "k_render_bwd": "k_render_bwd",
            "k_scatter": "k_scatter", "k_preprocess_bwd": "k_preprocess_bwd"}


def collect(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for pat, fam in FAMILIES.items():
            if pat in r["Kernel_Name"]:
                tot[fam][0] += 1
                tot[fam][1] += float(r["Counter_Value"])
                break
    return tot


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on "
               "`python3 bench.py --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace`; counters are KB; FETCH_SIZE "
               "doubled (gfx950 tallies the 128-B requests of 16 B/lane streams at 64 B, MI355X_MICROARCH.md HBM); "
               "WRITE_SIZE as read; per-launch averages over every launch of the kernel family",
       "_source_id": source_id()}      # the kernel sources this pass profiled: bench.py reports the traffic only for that build
for fam in sorted(set(fetch) | set(write)):
    n = max(fetch[fam][0], write[fam][0], 1)
    f_kb, w_kb = fetch[fam][1] / max(fetch[fam][0], 1), write[fam][1] / max(write[fam][0], 1)
    out[fam] = {"launches": n, "fetch_kb_raw": round(f_kb, 1), "write_kb": round(w_kb, 1),
                "hbm_bytes_per_launch": int((2.0 * f_kb + w_kb) * 1024)}
if len(sys.argv) > 4:
    fam = json.load(open(sys.argv[4]))
    gemm = [v for k, v in fam.items() if k.startswith("k_gemm") or k == "k_ffn320"]
    # MFMA busy of the whole contraction family: launch-weighted by the chip-active cycles of each member
    num = sum(v["per_launch"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * v["launches"] for v in gemm)
    den = sum(v["per_launch"].get("GRBM_GUI_ACTIVE", 0.0) * v["launches"] for v in gemm) / 8.0 * 1024.0
    if den > 0 and "k_gemm" in out:
        out["k_gemm"]["mfma_busy"] = round(num / den, 4)
    for name in ("k_attn_spatial", "k_render_bwd", "k_render"):
        if name in out and name in fam and "mfma_busy" in fam[name]:
            out[name]["mfma_busy"] = fam[name]["mfma_busy"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
