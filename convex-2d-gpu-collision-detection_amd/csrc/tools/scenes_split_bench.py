#!/usr/bin/env python3
"""Developer tool: the config-4 shard (4e6 data points, max_samples 120 000) as ONE c2d_mc_scenes call against the same
scenes cut into S sub-shards that run concurrently on S ctxs / streams (results are identical by construction: streams are
keyed by scene id).  usage: scenes_split_bench.py [n_scenes]"""
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


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    max_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 120_000
    engs = [pkg.Engine(0) for _ in range(4)]
    e0 = engs[0]
    tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
    d_p, d_s = e0.to_device(tp), e0.to_device(ts)
    d_sc = e0.empty(ns, pkg.SCENE_DT)
    e0.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, 0, ns, d_sc)
    d_h, d_u = e0.zeros(ns, np.uint32), e0.zeros(ns, np.uint32)
    e0.synchronize()
    streams = [e.stream_create() for e in engs]
    ref = None
    for S in (1, 2, 4, 1, 2, 4):
        e0.memset(d_h, 0, 4 * ns)
        e0.memset(d_u, 0, 4 * ns)
        e0.synchronize()
        t0 = time.perf_counter()
        for k in range(S):
            b, e = k * ns // S, (k + 1) * ns // S
            engs[k].mc_scenes_async(d_p, 65536, d_s, 65536, d_sc.ptr + b * pkg.SCENE_DT.itemsize, e - b, 4.07, 1.74, wl.DEFAULT_BINS,
                                    wl.DEFAULT_BIN_ACCURACY, max_samples, 11, b, d_h.ptr + 4 * b, d_u.ptr + 4 * b, stream=streams[k])
        for k in range(S):
            engs[k].synchronize(streams[k])
        dt = time.perf_counter() - t0
        u = d_u.get()
        h = d_h.get()
        tot = int(u.astype(np.int64).sum())
        if ref is None:
            ref = (u.copy(), h.copy())
        same = np.array_equal(u, ref[0]) and np.array_equal(h, ref[1])
        print(f"{S} concurrent sub-shard(s): {dt * 1e3:.1f} ms  {tot / dt / 1e9:.1f}e9 samples/s  identical to the single call: {same}")


if __name__ == "__main__":
    main()
