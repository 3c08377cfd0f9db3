#!/bin/bash
# Developer tool: copy the judged summaries of `bash tools/collect_profiles.sh <round>` from gpurun_out/<round>/ (scratch) into
# profiles/<round>/ (tracked).   usage: tools/copy_profiles.sh r04
set -e
cd "$(dirname "$0")/.."
r=$1; src=gpurun_out/$r; dst=profiles/$r
mkdir -p $dst/pmc
cp $src/bench.json $dst/bench_$r.json
cp $src/bench_kernel_stats.csv $src/traffic.json $src/unet_shapes.txt $src/norm_bench.txt $src/raster_breakdown.txt $src/merge_units.txt $src/lpips_bench.txt $dst/
cp $src/pair_profile.txt $dst/pair_profile.txt
cp $src/unet_shapes_f25.txt $src/shape_efficiency_f14.txt $src/shape_efficiency_f25.txt $src/trainer_breakdown.txt $dst/ 2>/dev/null || true
cp $src/pmc_counters_by_kernel.csv $src/pmc_counters_by_kernel.json $src/bench_FETCH_SIZE_by_kernel.csv $src/bench_WRITE_SIZE_by_kernel.csv $dst/pmc/
ls $dst $dst/pmc
