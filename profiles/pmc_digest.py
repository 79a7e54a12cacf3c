#!/usr/bin/env python3
"""Print per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv (c2d kernels only)."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "c2d::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:42s} {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
