#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/...) into the small summaries kept under profiles/.

usage: python profiles/summarize.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]
 - kernel stats: the --kernel-trace --stats CSV with kernel names cut to 90 characters
 - PMC: per-kernel mean FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) and the
   HBM bytes per launch after the gfx950 correction of MI355X_MICROARCH.md §HBM
   (FETCH_SIZE counts 128-B read requests as 64 B: double it; WRITE_SIZE is exact).
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name: str) -> str:
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    here = os.path.dirname(os.path.abspath(__file__))
    f = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    if len(sys.argv) >= 5:
        res = {}
        for which, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == which:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                res.setdefault(k, {})[which + "_KB_mean"] = sum(v) / len(v)
                res[k][which + "_launches"] = len(v)
        for k, v in res.items():
            fetch = v.get("FETCH_SIZE_KB_mean", 0.0) * 1024
            write = v.get("WRITE_SIZE_KB_mean", 0.0) * 1024
            v["hbm_bytes_per_launch_corrected"] = 2 * fetch + write
            v["correction"] = "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE reports half of streamed read bytes)"
        json.dump(res, open(os.path.join(here, f"{tag}_pmc_hbm.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
