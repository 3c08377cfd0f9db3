# usage: bash tools/pmc_one.sh <one_gemm kind> <kernel substring> "<counters>" ["<counters>" ...]   (developer tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
kind=$1; kern=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcone/p$i -- python3 $R/tools/one_gemm.py $kind > $R/gpurun_out/pmcone_$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $kern $(find $R/gpurun_out/pmcone -name "*counter_collection.csv")
f=$(find $R/gpurun_out/pmcone -name "*kernel_trace.csv" | head -1)
python3 -c "
import csv
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open('$f')) if '$kern' in r['Kernel_Name']]
print('dur us', sorted(d))"
rm -rf $R/gpurun_out/pmcone
