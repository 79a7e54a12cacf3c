#!/usr/bin/env python3
"""Print per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv.
usage: pmc_digest.py [--wide] [--all] DIR...     (default: c2d kernels only, names cut at 40 characters as the older digests;
--wide: 64 characters; --all: every kernel, e.g. the probe binary's)"""
import collections, csv, glob, os, sys
args = sys.argv[1:]
wide, every = "--wide" in args, "--all" in args
w = 64 if wide else 40
for d in (a for a in args if not a.startswith("--")):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if every or "c2d::" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:w], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:{w + 2}s} {c:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
