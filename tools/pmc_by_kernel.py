"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_by_kernel.py <dir with *_counter_collection.csv>."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = acc[(r["Kernel_Name"][:70], r["Grid_Size"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for (k, grid), cs in sorted(acc.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    print(k, "grid", grid, "launches", max(v[1] for v in cs.values()))
    print("   " + "  ".join(f"{c}={v[0] / v[1]:.4g}" for c, v in sorted(cs.items())))
