#!/usr/bin/env python3
"""Per-step trace of the adaptive Monte-Carlo loop on the reference-default batch (VERDICT r1
item 7): 1e5 scenes, max_samples 4 000 000 => 60 schedule steps (20 x 1000 samples — run as one launch —, then 40 x 100 000).

  scenes_trace.py run <out_dir>            runs the batch (twice: warm-up, then the traced run) and saves n_used;
                                            put it under `rocprofv3 --kernel-trace --output-format csv -d <trace_dir> -- python3 ...`
  scenes_trace.py digest <trace_dir> <out_dir> > table.md
                                            joins the kernel trace (start/end of every advance / decide launch of the LAST
                                            c2d_mc_scenes call) with the number of scenes still active at each step (from n_used)

Developer tool (GPU).  The question it answers: does the tail of the loop — few scenes left, 100 000 samples each —
keep the chip busy, and how much time do the launch gaps between the steps cost?"""
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
N_SCENES, N_TAB, MAX_SAMPLES = 100_000, 65536, 4_000_000


def schedule():
    steps, n = [], 0
    while n < MAX_SAMPLES:
        b = 1000 if n < 20000 else 100000
        steps.append((n, b))
        n += b
    return steps


def run(out_dir):
    from __graft_entry__ import load_package
    import importlib

    pkg = load_package()
    wl = importlib.import_module("c2d_amd.workloads")
    eng = pkg.Engine(0)
    tp, ts, _ = wl.random_tables(N_TAB, N_TAB, seed=7)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    d_sc = eng.empty(N_SCENES, pkg.SCENE_DT)
    eng.sample_scenes(d_p, N_TAB, d_s, N_TAB, 4.07, 1.74, 4.0, 7, 0, N_SCENES, d_sc)
    d_h, d_u = eng.zeros(N_SCENES, np.uint32), eng.zeros(N_SCENES, np.uint32)
    import time

    for rep in range(2):
        t0 = time.perf_counter()
        total, iters = eng.mc_scenes(d_p, N_TAB, d_s, N_TAB, d_sc, N_SCENES, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, MAX_SAMPLES,
                                     11, 0, d_h, d_u, None)
        dt = time.perf_counter() - t0
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "n_used.npy"), d_u.get())
    open(os.path.join(out_dir, "run.txt"), "w").write(f"total_samples {total} steps {iters} host_seconds {dt:.4f}\n")
    print(f"total samples {total}, {iters} steps, {dt * 1e3:.1f} ms (host clock, under the profiler)")


def digest(trace_dir, out_dir):
    f = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "mc_scenes_advance_kernel" in r["Kernel_Name"] or "mc_scenes_decide_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = schedule()
    # the leading small-batch steps (batch <= 1024 samples) run as ONE advance / decide pair (the "burst", c2d_mc.hip)
    burst = sum(1 for n0, b in steps if b <= 1024 and n0 < 20000)
    burst = burst if burst > 1 else 0
    groups = ([steps[:burst]] if burst else []) + [[st] for st in steps[burst:]]
    rows = rows[-2 * len(groups):]                     # the last call
    used = np.load(os.path.join(out_dir, "n_used.npy")).astype(np.int64)
    print(open(os.path.join(out_dir, "run.txt")).read().strip())
    print()
    print("| steps | samples before | batch | active scenes at the first step | advance us | decide us | gaps us | samples drawn | 1e9 samples/s |")
    print("|---|---|---|---|---|---|---|---|---|")
    t_first = int(rows[0]["Start_Timestamp"])
    prev_end = None
    tot_adv = tot_dec = tot_gap = 0.0
    for i, grp in enumerate(groups):
        adv, dec = rows[2 * i], rows[2 * i + 1]
        assert "advance" in adv["Kernel_Name"] and "decide" in dec["Kernel_Name"]
        a_us = (int(adv["End_Timestamp"]) - int(adv["Start_Timestamp"])) / 1e3
        d_us = (int(dec["End_Timestamp"]) - int(dec["Start_Timestamp"])) / 1e3
        gap = 0.0 if prev_end is None else (int(adv["Start_Timestamp"]) - prev_end) / 1e3
        gap2 = (int(dec["Start_Timestamp"]) - int(adv["End_Timestamp"])) / 1e3
        prev_end = int(dec["End_Timestamp"])
        n0, b = grp[0]
        active = int((used > n0).sum())
        samples = sum(int((used > m0).sum()) * bb for m0, bb in grp)
        tot_adv += a_us
        tot_dec += d_us
        tot_gap += gap + gap2
        label = f"{i if not burst else (0 if i == 0 else burst + i - 1)}" if len(grp) == 1 else f"0-{len(grp) - 1} (one launch)"
        print(f"| {label} | {n0} | {b} | {active} | {a_us:.1f} | {d_us:.1f} | {gap + gap2:.1f} | {samples} | {samples / a_us / 1e3 if a_us > 0 else 0:.1f} |")
    span = (prev_end - t_first) / 1e3
    print()
    print(f"first advance start .. last decide end: {span:.1f} us; advance kernels {tot_adv:.1f} us ({100 * tot_adv / span:.1f} %), "
          f"decide kernels {tot_dec:.1f} us ({100 * tot_dec / span:.1f} %), gaps between kernels {tot_gap:.1f} us ({100 * tot_gap / span:.1f} %)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        digest(sys.argv[2], sys.argv[3])
