"""Per-kernel-family summary of rocprofv3 --pmc passes (SQ / GRBM counters) -> JSON + CSV.

usage: python tools/pmc_families.py <out.json> <out.csv> <counter_collection.csv> [<counter_collection.csv> ...]

Families are matched on the kernel name (first match wins).  Derived figures:
  mfma_busy   = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 * 1024)     rocprofiler-sdk's MfmaUtil for gfx950:
                busy cycles summed over the 1024 SIMDs / (active cycles of the chip x SIMD count); GRBM_GUI_ACTIVE is reported
                summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
  valu_per_mfma_mop, lds_conflict_frac (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE-or-ACTIVE_INST_LDS) where the counters exist.
"""
import collections
import csv
import json
import sys

FAMILIES = [("k_gemm_g256", "k_gemm_g256"), ("k_gemm_dmapd", "k_gemm_dmapd"), ("k_gemm_widep", "k_gemm_widep"), ("k_gemm_z", "k_gemm_z"), ("k_gemm_dmap", "k_gemm_dmap"), ("k_gemm_dma<1, 256>", "k_gemm_dma<1,256>"),
            ("k_gemm_dma<1,256>", "k_gemm_dma<1,256>"), ("k_gemm_dma", "k_gemm_dma<other>"), ("k_gemm_w128", "k_gemm_w128"),
            ("k_ffn320", "k_ffn320"), ("k_lnlin320", "k_lnlin320"), ("k_gemm_skinny", "k_gemm_skinny"), ("k_attn_spatial", "k_attn_spatial"),
            ("k_attn_temporal", "k_attn_temporal"), ("k_render_bwd", "k_render_bwd"), ("k_render", "k_render"),
            ("k_gn_", "k_gn_*"), ("k_layernorm", "k_layernorm")]


def family(name):
    for pat, fam in FAMILIES:
        if pat in name:
            return fam
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[3:]:
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        c = agg[fam][r["Counter_Name"]]
        c[0] += 1
        c[1] += float(r["Counter_Value"])
out = {}
for fam, ctrs in sorted(agg.items()):
    e = {"launches": min(v[0] for v in ctrs.values()), "per_launch": {k: round(v[1] / v[0], 1) for k, v in sorted(ctrs.items())}}
    g = ctrs.get("GRBM_GUI_ACTIVE")
    m = ctrs.get("SQ_VALU_MFMA_BUSY_CYCLES")
    avg = lambda c: c[1] / c[0]                     # per launch (GRBM_GUI_ACTIVE is collected in every pass, the others in one)
    if g and m and g[1] > 0:
        e["mfma_busy"] = round(avg(m) / (avg(g) / 8.0 * 1024.0), 4)
    if g and "SQ_BUSY_CYCLES" in ctrs:
        e["sq_busy_over_gui_active"] = round(avg(ctrs["SQ_BUSY_CYCLES"]) / avg(g), 4)
    iv, im = ctrs.get("SQ_INSTS_VALU"), ctrs.get("SQ_INSTS_VALU_MFMA_MOPS_F16")
    if iv and im and im[1] > 0:
        e["valu_insts_per_mfma_mop_f16"] = round(avg(iv) / avg(im), 4)
    bc = ctrs.get("SQ_LDS_BANK_CONFLICT")
    la = ctrs.get("SQ_LDS_IDX_ACTIVE") or ctrs.get("SQ_ACTIVE_INST_LDS")
    if bc and la and la[1] > 0:
        e["lds_bank_conflict_over_lds_active"] = round(avg(bc) / avg(la), 4)
    out[fam] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
with open(sys.argv[2], "w") as f:
    names = sorted({c for e in out.values() for c in e["per_launch"]})
    f.write("family,launches,mfma_busy," + ",".join(names) + "\n")
    for fam, e in out.items():
        f.write(f'"{fam}",{e["launches"]},{e.get("mfma_busy", "")},' + ",".join(str(e["per_launch"].get(c, "")) for c in names) + "\n")
print(json.dumps(out, indent=1))
