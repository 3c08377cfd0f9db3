"""Sum rocprofv3 --pmc counter_collection.csv rows per (kernel, counter) (developer tool)."""
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sys.argv[1] in r["Kernel_Name"]:
            k = r["Counter_Name"]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:32s} launches {n:4d}  per-launch {v / n:16.1f}")
