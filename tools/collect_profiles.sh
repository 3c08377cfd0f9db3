#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r01
# Writes gpurun_out/<round>/ : bench line, rocprofv3 kernel stats of the same command, FETCH/WRITE passes.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r01}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 "$R/bench.py" --steps 3 --warmup 1 $BENCH_EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[profiles] bench done"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-sub-benchmarks > "$OUT/stats.log" 2>&1
echo "[profiles] kernel stats done"
timeout -k 10 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/fetch.log" 2>&1 || { echo "[profiles] FETCH_SIZE pass FAILED:"; tail -20 "$OUT/fetch.log"; exit 3; }
echo "[profiles] FETCH_SIZE pass done"
timeout -k 10 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/write.log" 2>&1 || { echo "[profiles] WRITE_SIZE pass FAILED:"; tail -20 "$OUT/write.log"; exit 3; }
echo "[profiles] WRITE_SIZE pass done"
# SQ / GRBM passes: matrix-pipe busy cycles, instruction mix, LDS bank conflicts per kernel family.
# Counter sets are intersected with what `rocprofv3 -L` lists on this box; each pass is its own run (program directly after --).
AVAIL=$(rocprofv3 -L 2>&1 || true)
pick() { local out=""; for c in "$@"; do if grep -qw "$c" <<< "$AVAIL"; then out="$out $c"; fi; done; echo $out; }
SET_A=$(pick SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE)
SET_B=$(pick SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE)
echo "[profiles] SQ pass A:$SET_A" ; echo "[profiles] SQ pass B:$SET_B"
timeout -k 10 150 rocprofv3 --kernel-trace --pmc $SET_A --output-format csv -d "$OUT/sqa" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/sqa.log" 2>&1 || { echo "[profiles] SQ pass A FAILED:"; tail -20 "$OUT/sqa.log"; }
timeout -k 10 150 rocprofv3 --kernel-trace --pmc $SET_B --output-format csv -d "$OUT/sqb" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/sqb.log" 2>&1 || { echo "[profiles] SQ pass B FAILED:"; tail -20 "$OUT/sqb.log"; }
SQ_FILES=$(ls "$OUT"/sqa/p_counter_collection.csv "$OUT"/sqb/p_counter_collection.csv 2>/dev/null || true)
if [ -n "$SQ_FILES" ]; then
  python3 "$R/tools/pmc_families.py" "$OUT/pmc_counters_by_kernel.json" "$OUT/pmc_counters_by_kernel.csv" $SQ_FILES > /dev/null
  echo "[profiles] SQ passes summarised"
fi
cp "$OUT/stats/bench_kernel_stats.csv" "$OUT/bench_kernel_stats.csv"
( timeout -k 10 200 python3 "$R/tools/unet_breakdown.py" 14 detail 2>/dev/null | grep -v amdgpu.ids > "$OUT/unet_shapes.txt" ) || true
( timeout -k 10 200 python3 "$R/tools/unet_breakdown.py" 25 detail 2>/dev/null | grep -v amdgpu.ids > "$OUT/unet_shapes_f25.txt" ) || true
# per-shape TFLOP/s of the SAME build (the tables used to predate the final kernels)
python3 "$R/tools/shape_efficiency.py" "$OUT/unet_shapes.txt" > "$OUT/shape_efficiency_f14.txt" || true
python3 "$R/tools/shape_efficiency.py" "$OUT/unet_shapes_f25.txt" > "$OUT/shape_efficiency_f25.txt" || true
if [ -f "$OUT/fetch/p_counter_collection.csv" ] && [ -f "$OUT/write/p_counter_collection.csv" ]; then
  python3 "$R/tools/traffic_json.py" "$OUT/fetch/p_counter_collection.csv" "$OUT/write/p_counter_collection.csv" "$OUT/traffic.json" $( [ -f "$OUT/pmc_counters_by_kernel.json" ] && echo "$OUT/pmc_counters_by_kernel.json" ) $( [ -s "$OUT/unet_shapes.txt" ] && echo "$OUT/unet_shapes.txt" ) > /dev/null
  python3 "$R/tools/pmc_by_kernel.py" "$OUT/fetch/p_counter_collection.csv" FETCH_SIZE > "$OUT/bench_FETCH_SIZE_by_kernel.csv"
  python3 "$R/tools/pmc_by_kernel.py" "$OUT/write/p_counter_collection.csv" WRITE_SIZE > "$OUT/bench_WRITE_SIZE_by_kernel.csv"
fi
# per-shape / per-kernel breakdowns of the same build (developer tools; failures here do not fail the collection)
( timeout -k 10 100 python3 "$R/tools/raster_breakdown.py" 2>/dev/null | grep -v amdgpu.ids > "$OUT/raster_breakdown.txt" ) || true
( timeout -k 10 150 python3 "$R/tools/trainer_breakdown.py" 20 2>/dev/null | grep -v amdgpu.ids > "$OUT/trainer_breakdown.txt" ) || true
( timeout -k 10 200 python3 "$R/tools/merge_units.py" 25 2>/dev/null | grep -v amdgpu.ids > "$OUT/merge_units.txt" ) || true
( timeout -k 10 100 python3 "$R/tools/norm_bench.py" 2>/dev/null | grep -v amdgpu.ids > "$OUT/norm_bench.txt" ) || true
( timeout -k 10 100 python3 "$R/tools/lpips_bench.py" 2>/dev/null | grep -v amdgpu.ids > "$OUT/lpips_bench.txt" ) || true
( timeout -k 10 200 python3 "$R/tools/pair_profile.py" 2>/dev/null | grep -v amdgpu.ids | head -40 > "$OUT/pair_profile.txt" ) || true
rm -rf "$OUT/fetch/p_kernel_trace.csv" "$OUT/write/p_kernel_trace.csv" "$OUT/stats/bench_kernel_trace.csv" "$OUT/sqa/p_kernel_trace.csv" "$OUT/sqb/p_kernel_trace.csv"
# the raw per-dispatch counter files are large: keep the per-kernel summaries only
rm -f "$OUT"/sqa/p_counter_collection.csv "$OUT"/sqb/p_counter_collection.csv "$OUT"/fetch/p_counter_collection.csv "$OUT"/write/p_counter_collection.csv
cat "$OUT/bench.json"
