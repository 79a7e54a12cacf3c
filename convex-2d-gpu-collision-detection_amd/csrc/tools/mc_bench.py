#!/usr/bin/env python3
"""Developer tool: wall time of the three Monte-Carlo workloads the docs quote, through the C-ABI —
  config 3   one scene, 1e8 samples (c2d_mc_pair, the bench scene);
  config 4   the per-GPU shard: 4e6 data points, max_samples 120 000 (c2d_mc_scenes);
  default    the reference-default batch: 1e5 data points, max_samples 4 020 000.
usage: mc_bench.py [--pair-only] [lib.so ...]    (default: lib/libc2d.so; several libraries = an A/B of builds on the same box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import importlib  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")


def scenes(e0, ns, max_samples, reps=3):
    tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
    d_p, d_s = e0.to_device(tp), e0.to_device(ts)
    d_sc = e0.empty(ns, pkg.SCENE_DT)
    e0.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, 0, ns, d_sc)
    d_h, d_u = e0.zeros(ns, np.uint32), e0.zeros(ns, np.uint32)
    e0.synchronize()
    st = e0.stream_create()
    best = 1e9
    for _ in range(reps):
        e0.memset(d_h, 0, 4 * ns)
        e0.memset(d_u, 0, 4 * ns)
        e0.synchronize()
        t0 = time.perf_counter()
        e0.mc_scenes_async(d_p, 65536, d_s, 65536, d_sc.ptr, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, max_samples, 11, 0,
                           d_h.ptr, d_u.ptr, stream=st)
        e0.synchronize(st)
        best = min(best, time.perf_counter() - t0)
    tot = int(d_u.get().astype(np.int64).sum())
    hits = int(d_h.get().astype(np.int64).sum())
    print(f"  {ns} data points, max_samples {max_samples}: {best * 1e3:.1f} ms, {tot / best / 1e9:.1f}e9 samples/s  (samples {tot}, hits {hits})", flush=True)
    for d in (d_p, d_s, d_sc, d_h, d_u):
        d.free()


def pair(e0, n=100_000_000, reps=20):
    sc = wl.MC_PAIR_SCENE
    d = e0.zeros(1, np.uint64)
    best = 1e9
    for _ in range(4):  # `reps` calls back to back, one synchronisation: the launch overhead overlaps
        e0.memset(d, 0, 8)
        e0.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e0.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, n, d)
        e0.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print(f"  one scene, {n} samples: {best * 1e3:.3f} ms, {n / best / 1e9:.1f}e9 samples/s  (hits {int(d.get()[0]) // reps})", flush=True)
    d.free()


def main():
    pair_only = "--pair-only" in sys.argv
    libs = [a for a in sys.argv[1:] if not a.startswith("--")] or [None]
    for lib in libs:
        print(os.path.basename(lib) if lib else "lib/libc2d.so", flush=True)
        e0 = pkg.Engine(0, lib_path=lib)
        pair(e0)
        if not pair_only:
            scenes(e0, 4_000_000, 120_000)
            scenes(e0, 100_000, 4_020_000)
        e0.close()


if __name__ == "__main__":
    main()
