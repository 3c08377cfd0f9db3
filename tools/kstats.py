"""Print a rocprofv3 kernel_stats.csv compactly (developer tool): python tools/kstats.py file.csv [divisor]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "").split("(")[0][:64]
    print(f"{n:66s} {int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:9.1f} us  total/{div:g} {int(r['TotalDurationNs']) / 1e3 / div:10.1f} us")
