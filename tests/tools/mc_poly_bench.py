#!/usr/bin/env python3
"""Developer tool: throughput of c2d_mc_poly_pair on single scenes — the bench scene of the polygon leg at several vertex counts,
rectangles given as 4-gons next to c2d_mc_pair on the same scene, a far scene — and of c2d_mc_poly_scenes on a random dataset
("Monte-Carlo over polygons").
usage: mc_poly_bench.py [lib.so ...]   (several libraries = an A/B of builds in one process)"""
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
W, H = 4.07, 1.74


def timed(eng, st, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        eng.synchronize()
        t0 = time.perf_counter()
        fn()
        eng.synchronize(st)
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    libs = sys.argv[1:] or [pkg.library_path()]
    n = int(os.environ.get("N", 400_000_000))
    for lib in libs:
        eng = pkg.Engine(0, lib_path=os.path.abspath(lib))
        st = eng.stream_create()
        print(os.path.basename(lib), flush=True)
        d = eng.zeros(1, np.uint64)
        cases = [("bench scene %d x %d" % (ka, kb), wl.mc_poly_pair_scene(ka, kb)) for ka, kb in [(7, 5), (3, 3), (4, 4), (8, 8), (12, 12), (16, 16), (16, 4), (4, 16)]]
        far = wl.mc_poly_pair_scene(7, 5)
        far["pos"] = (9.0, 1.0)
        far["std_dev"] = (0.8, 0.8, 0.2, 0.0, 0.0)
        cases.append(("far scene 7 x 5 (radius word)", far))
        spread = wl.mc_poly_pair_scene(7, 5)
        spread["std_dev"] = (3.0, 3.0, 0.2, 0.0, 0.0)
        cases.append(("sigma 3: most samples ruled out by their centre", spread))
        shape = wl.mc_poly_pair_scene(7, 5)
        shape["std_dev"] = (0.3, 0.3, 0.2, 0.1, 0.1)
        cases.append(("bench scene 7 x 5 with shape noise", shape))
        for name, sc in cases:
            eng.memset(d.ptr, 0, 8)
            t = timed(eng, st, lambda: eng.mc_poly_pair(sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1234, 0, 0, n, d, stream=st))
            hits = int(d.get()[0]) // 3
            print(f"  {name}: {n / t / 1e9:.2f}e9 samples/s ({t * 1e3:.2f} ms), p = {hits / n:.5f}", flush=True)
        # rectangles both ways
        rs = wl.MC_PAIR_SCENE
        eng.memset(d.ptr, 0, 8)
        t = timed(eng, st, lambda: eng.mc_pair(W, H, rs["pos"], rs["pose"], rs["std_dev"], 1234, 0, 0, n, d, stream=st))
        print(f"  config-3 scene, c2d_mc_pair: {n / t / 1e9:.2f}e9 samples/s", flush=True)
        t = timed(eng, st, lambda: eng.mc_poly_pair(wl.rect_polygon(W, H), rs["pos"], rs["pose"][2], wl.rect_polygon(2.0, 1.0), rs["std_dev"], 1234, 0, 0, n, d, stream=st))
        print(f"  config-3 scene as 4-gons, c2d_mc_poly_pair: {n / t / 1e9:.2f}e9 samples/s", flush=True)
        # adaptive dataset
        ns = int(os.environ.get("SCENES", 400_000))
        poses, sds = wl.random_poly_tables(4096, 4096, seed=7)
        scenes = wl.random_poly_scenes(ns, poses, sds, 2.3, seed=8)
        robot = wl.mc_poly_pair_scene(9, 5)["robot"]
        d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
        d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
        res = {}

        def run():
            res["r"] = eng.mc_poly_scenes(robot, d_p, len(poses), d_s, len(sds), d_sc, ns, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11, 0, d_h, d_u, None, stream=st)

        t = timed(eng, st, run, reps=2)
        total, iters = res["r"]
        u = d_u.get()
        print(f"  adaptive: {ns} polygon scenes in {t * 1e3:.1f} ms = {ns / t / 1e6:.2f}e6 scenes/s, {total / t / 1e9:.2f}e9 drawn samples/s, {iters} steps, "
              f"mean {total / ns:.0f} samples per scene, {np.mean(u >= 120_000) * 100:.0f} % at the cap", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
