"""Summarise rocprofv3 counter_collection CSVs: per kernel name, mean of each counter per dispatch."""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "")
            if "conv_" not in k:
                continue
            short = k.split("(")[0][-60:]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
