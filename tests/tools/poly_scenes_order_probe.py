#!/usr/bin/env python3
"""Developer tool: does the ORDER of the scenes matter to c2d_mc_poly_scenes?  The polygon evaluation is instantiated per obstacle
vertex count (sixteen instances), and the waves of a CU work on neighbouring scenes: with the scenes in random order they run
different instances side by side (instruction cache, divergent code paths), with the scenes sorted by vertex count they share one.
The same multiset of scenes is run as generated, sorted by the obstacle's vertex count, and sorted by (vertex count, pose index).
usage: poly_scenes_order_probe.py [lib.so]      SCENES=<n> (default 400000)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else pkg.library_path()
    ns = int(os.environ.get("SCENES", 400_000))
    eng = pkg.Engine(0, lib_path=os.path.abspath(lib))
    st = eng.stream_create()
    poses, sds = wl.random_poly_tables(4096, 4096, seed=7)
    scenes = wl.random_poly_scenes(ns, poses, sds, 2.3, seed=8)
    robot = wl.mc_poly_pair_scene(9, 5)["robot"]
    kb = poses["obstacle"]["k"][scenes["pose_idx"].astype(np.int64)]
    orders = {"as generated": np.arange(ns),
              "sorted by the obstacle's vertex count": np.argsort(kb, kind="stable"),
              "sorted by vertex count, then pose": np.lexsort((scenes["pose_idx"], kb)),
              "as generated, again": np.arange(ns)}
    d_p, d_s = eng.to_device(poses), eng.to_device(sds)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    print(f"{os.path.basename(lib)}: {ns} scenes, vertex counts {np.bincount(kb, minlength=17)[1:].tolist()}", flush=True)
    for name, order in orders.items():
        d_sc = eng.to_device(np.ascontiguousarray(scenes[order]))
        best, res = 1e9, None
        for _ in range(3):
            eng.memset(d_h.ptr, 0, 4 * ns)
            eng.memset(d_u.ptr, 0, 4 * ns)
            eng.synchronize()
            t0 = time.perf_counter()
            res = eng.mc_poly_scenes(robot, d_p, len(poses), d_s, len(sds), d_sc, ns, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11, 0, d_h, d_u, None, stream=st)
            eng.synchronize(st)
            best = min(best, time.perf_counter() - t0)
        total, iters = res
        print(f"  {name:40s} {best * 1e3:8.1f} ms  {ns / best / 1e6:6.2f}e6 scenes/s  {total / best / 1e9:7.1f}e9 drawn samples/s  ({total} samples, {iters} steps)", flush=True)
        d_sc.free()
    eng.close()


if __name__ == "__main__":
    main()
