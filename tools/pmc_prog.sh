# usage: bash tools/pmc_prog.sh <tools/script.py> <kernel substring> "<counters>" ["<counters>" ...]   (developer tool)
# One rocprofv3 --pmc pass per counter group over `python3 <script>`, then per-launch sums for the named kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prog=$1; kern=$2; shift 2
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcprog/p$i -- python3 $R/$prog > $R/gpurun_out/pmcprog_$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $kern $(find $R/gpurun_out/pmcprog -name "*counter_collection.csv")
f=$(find $R/gpurun_out/pmcprog -name "*kernel_trace.csv" | head -1)
python3 -c "
import csv
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open('$f')) if '$kern' in r['Kernel_Name']]
print('dur us', sorted(d))"
rm -rf $R/gpurun_out/pmcprog
