"""Sum one rocprofv3 --pmc counter per kernel name (developer tool): python tools/pmc_by_kernel.py <counter_collection.csv> <COUNTER>"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print("kernel,launches,sum_KB,per_launch_KB")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{k}",{n},{v:.1f},{v / n:.1f}')
