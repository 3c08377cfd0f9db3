#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r01
# Writes gpurun_out/<round>/ : bench line, rocprofv3 kernel stats of the same command, FETCH/WRITE passes.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r01}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 python3 "$R/bench.py" --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[profiles] bench done"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-sub-benchmarks > "$OUT/stats.log" 2>&1
echo "[profiles] kernel stats done"
timeout -k 10 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/fetch.log" 2>&1 || { echo "[profiles] FETCH_SIZE pass FAILED:"; tail -20 "$OUT/fetch.log"; exit 3; }
echo "[profiles] FETCH_SIZE pass done"
timeout -k 10 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- python3 "$R/bench.py" --steps 1 --warmup 0 --raster-iters 5 --no-cpu-baseline --no-sub-benchmarks --no-kernel-trace > "$OUT/write.log" 2>&1 || { echo "[profiles] WRITE_SIZE pass FAILED:"; tail -20 "$OUT/write.log"; exit 3; }
echo "[profiles] WRITE_SIZE pass done"
cp "$OUT/stats/bench_kernel_stats.csv" "$OUT/bench_kernel_stats.csv"
if [ -f "$OUT/fetch/p_counter_collection.csv" ] && [ -f "$OUT/write/p_counter_collection.csv" ]; then
  python3 "$R/tools/traffic_json.py" "$OUT/fetch/p_counter_collection.csv" "$OUT/write/p_counter_collection.csv" "$OUT/traffic.json" > /dev/null
  python3 "$R/tools/pmc_by_kernel.py" "$OUT/fetch/p_counter_collection.csv" FETCH_SIZE > "$OUT/bench_FETCH_SIZE_by_kernel.csv"
  python3 "$R/tools/pmc_by_kernel.py" "$OUT/write/p_counter_collection.csv" WRITE_SIZE > "$OUT/bench_WRITE_SIZE_by_kernel.csv"
fi
# per-shape / per-kernel breakdowns of the same build (developer tools; failures here do not fail the collection)
( timeout -k 10 200 python3 "$R/tools/unet_breakdown.py" 14 detail 2>/dev/null | grep -v amdgpu.ids > "$OUT/unet_shapes.txt" ) || true
( timeout -k 10 100 python3 "$R/tools/raster_breakdown.py" 2>/dev/null | grep -v amdgpu.ids > "$OUT/raster_breakdown.txt" ) || true
rm -rf "$OUT/fetch/p_kernel_trace.csv" "$OUT/write/p_kernel_trace.csv" "$OUT/stats/bench_kernel_trace.csv"
cat "$OUT/bench.json"
