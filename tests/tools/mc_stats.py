#!/usr/bin/env python3
"""Developer tool: where the Monte-Carlo samples of the bench workloads go, counted by the census build of the kernels
(`make lib-mcstats`, -DC2D_MC_STATS): far / near path, candidates left by the radius-word test, centres evaluated, samples
that reach the full evaluation, evaluation passes that the closed-form test could not decide, hits.  With `--record` the
evaluated-sample fraction of the config-4 shard is written into profiles/measured_counts.json (bench.py quotes it beside the
drawn-sample rate).  usage: mc_stats.py [--record]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")
LIB = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "lib", "libc2d_mcstats.so")
NAMES = ["samples", "far_path", "near_path", "radius_candidates", "centres_evaluated", "fully_evaluated", "parallel_axis_fallbacks", "hits",
         "vertex_arithmetic_passes", "unused9", "unused10", "unused11"]


def stats(eng, reset=True):
    out = (C.c_ulonglong * 12)()
    fn = eng.lib.c2d_debug_mc_stats
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    assert fn(eng.h, out, 1 if reset else 0) == 0
    return dict(zip(NAMES, [int(v) for v in out]))


def show(title, st):
    n = st["samples"]
    print(f"{title}: {n:.4g} samples; far path {st['far_path'] / n:.3f}, near path {st['near_path'] / n:.3f}; radius candidates (far) "
          f"{st['radius_candidates'] / max(st['far_path'], 1):.5f} of the far samples; centres evaluated {st['centres_evaluated'] / n:.4f}; "
          f"fully evaluated {st['fully_evaluated'] / n:.5f}; hits {st['hits'] / n:.5f}; evaluation passes (64 lanes) in which a thin closed-form "
          f"result sent the wave through the vertex arithmetic: {st['vertex_arithmetic_passes']} of about {st['fully_evaluated'] / 64:.4g} "
          f"(and {st['parallel_axis_fallbacks']} second axes of a parallel pair evaluated in those)")


def scenes(eng, ns, max_samples):
    tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    stats(eng)
    eng.mc_scenes(d_p, 65536, d_s, 65536, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, max_samples, 11, 0, d_h, d_u, None)
    st = stats(eng)
    assert st["hits"] == int(d_h.get().astype(np.int64).sum()) and st["samples"] == int(d_u.get().astype(np.int64).sum())
    for a in (d_p, d_s, d_sc, d_h, d_u):
        a.free()
    return st


POLY_NAMES = ["samples", "far_path", "near_path", "radius_candidates", "centres_evaluated", "evaluated", "survive_robot_normals_queued", "hits",
              "obstacle_normals_in_place", "obstacle_normals_from_queue", "unused10", "unused11"]


def poly_stats(eng, reset=True):
    out = (C.c_ulonglong * 12)()
    fn = eng.lib.c2d_debug_mc_poly_stats
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    assert fn(eng.h, out, 1 if reset else 0) == 0
    return dict(zip(POLY_NAMES, [int(v) for v in out]))


def poly_show(title, st):
    n, ev = st["samples"], max(st["evaluated"], 1)
    print(f"{title}: {n:.4g} samples; far path {st['far_path'] / n:.3f}, near path {st['near_path'] / n:.3f}; radius candidates (far) "
          f"{st['radius_candidates'] / max(st['far_path'], 1):.5f} of the far samples; centres evaluated {st['centres_evaluated'] / n:.4f}; "
          f"reach the evaluation {st['evaluated'] / n:.5f}; hits {st['hits'] / n:.5f} (= {st['hits'] / ev:.3f} of the evaluated); of the evaluated: "
          f"obstacle normals in place {st['obstacle_normals_in_place'] / ev:.3f}, survivors of the robot's normals queued {st['survive_robot_normals_queued'] / ev:.3f} "
          f"(evaluated from the queue {st['obstacle_normals_from_queue'] / ev:.3f}), decided by the robot's normals alone "
          f"{1 - (st['obstacle_normals_in_place'] + st['survive_robot_normals_queued']) / ev:.3f}")


def poly_main(eng):
    ps = wl.mc_poly_pair_scene()
    d = eng.zeros(1, np.uint64)
    poly_stats(eng)
    eng.mc_poly_pair(ps["robot"], ps["pos"], ps["theta"], ps["obstacle"], ps["std_dev"], 1234, 0, 0, 100_000_000, d)
    st = poly_stats(eng)
    assert st["hits"] == int(d.get()[0])
    poly_show("polygon bench scene (7-gon against pentagon, 1e8 samples)", st)
    ns = 200_000
    poses, sds = wl.random_poly_tables(4096, 4096, seed=7)
    scn = wl.random_poly_scenes(ns, poses, sds, 2.3, seed=8)
    robot = wl.mc_poly_pair_scene(9, 5)["robot"]
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scn)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    poly_stats(eng)
    eng.mc_poly_scenes(robot, d_p, len(poses), d_s, len(sds), d_sc, ns, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11, 0, d_h, d_u, None)
    st = poly_stats(eng)
    assert st["hits"] == int(d_h.get().astype(np.int64).sum()) and st["samples"] == int(d_u.get().astype(np.int64).sum())
    poly_show("adaptive polygon dataset of the bench (2e5 scenes, max_samples 120 000)", st)


def main():
    eng = pkg.Engine(0, lib_path=LIB)
    sc = wl.MC_PAIR_SCENE
    d = eng.zeros(1, np.uint64)
    stats(eng)
    eng.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 100_000_000, d)
    show("config 3 (one scene, 1e8 samples)", stats(eng))
    st4 = scenes(eng, 4_000_000, 120_000)
    show("config 4 shard (4e6 data points, max_samples 120 000)", st4)
    show("reference-default batch (1e5 data points, max_samples 4 020 000)", scenes(eng, 100_000, 4_020_000))
    poly_main(eng)
    if "--record" in sys.argv:
        path = os.path.join(ROOT, "profiles", "measured_counts.json")
        cur = json.load(open(path))
        e = cur.get("mc_scenes.config4")
        if e:
            e["evaluated_fraction"] = round(st4["fully_evaluated"] / st4["samples"], 6)
            e["evaluated_source"] = ("tests/tools/mc_stats.py on the census build (make lib-mcstats): %d of %d drawn samples of the shard reach the "
                                     "full evaluation, %d get their centre evaluated" % (st4["fully_evaluated"], st4["samples"], st4["centres_evaluated"]))
            json.dump(cur, open(path, "w"), indent=1)
            print("recorded evaluated_fraction", e["evaluated_fraction"])


if __name__ == "__main__":
    main()
