#!/usr/bin/env python3
"""Developer tool: the clock the Monte-Carlo kernels HOLD, from the clock build of the library (`make lib-mcclock`, -DC2D_MC_CLOCK:
every wave stamps s_memtime and s_memrealtime around its sample work; the sums of the differences give the time-weighted shader
clock).  bench.py prices the VALU rooflines of its Monte-Carlo legs at the nominal 2.4 GHz AND at this clock.  With `--record` the
clocks of the config-3 scene and of the config-4 shard are written into profiles/measured_counts.json.
usage: mc_clock.py [--record] [tag for the source note]"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
LIB = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "lib", "libc2d_mcclock.so")


def clock(eng, which, reset=True, poly=False):
    out = (C.c_ulonglong * 4)()
    fn = eng.lib.c2d_debug_mc_poly_clock if poly else eng.lib.c2d_debug_mc_clock
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong), C.c_int]
    assert fn(eng.h, which, out, 1 if reset else 0) == 0
    cycles, ticks, waves = int(out[0]), int(out[1]), int(out[2])
    return {"shader_cycles": cycles, "ticks_100MHz": ticks, "waves": waves, "ghz": (cycles / ticks * 0.1) if ticks else float("nan")}


def scenes(eng, ns, max_samples):
    tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    eng.mc_scenes(d_p, 65536, d_s, 65536, d_sc, min(ns, 100_000), 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, max_samples, 11, 0, d_h, d_u, None)  # warm
    clock(eng, 1)
    eng.mc_scenes(d_p, 65536, d_s, 65536, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, max_samples, 11, 0, d_h, d_u, None)
    c = clock(eng, 1)
    for a in (d_p, d_s, d_sc, d_h, d_u):
        a.free()
    return c


def poly_scenes(eng, n):
    """bench.py's adaptive polygon sub-leg (mc_poly.scenes)"""
    pp, ps = wl.random_poly_tables(4096, 4096, seed=7)
    scn = wl.random_poly_scenes(n, pp, ps, 2.3, seed=8)
    rob = wl.mc_poly_pair_scene(9, 5)["robot"]
    d_pp, d_ps, d_sc = eng.to_device(pp), eng.to_device(ps), eng.to_device(scn)
    d_h, d_u = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32)
    run = lambda: eng.mc_poly_scenes(rob, d_pp, len(pp), d_ps, len(ps), d_sc, n, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120_000, 11, 0, d_h, d_u, None)  # noqa: E731
    run()  # warm
    clock(eng, 1, poly=True)
    run()
    c = clock(eng, 1, poly=True)
    for a in (d_pp, d_ps, d_sc, d_h, d_u):
        a.free()
    return c


def main():
    eng = pkg.Engine(0, lib_path=LIB)
    sc = wl.MC_PAIR_SCENE
    d = eng.zeros(1, np.uint64)
    for _ in range(30):  # warm: the clocks ramp for ~15 ms after an idle period
        eng.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 100_000_000, d)
    eng.synchronize()
    clock(eng, 0)
    for _ in range(20):
        eng.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 100_000_000, d)
    c3 = clock(eng, 0)
    print(f"config 3 (mc_pair_kernel, 20 x 1e8 samples back to back): {c3['waves']} waves, {c3['shader_cycles']:.4g} shader cycles in "
          f"{c3['ticks_100MHz']:.4g} ticks of 100 MHz: the waves held {c3['ghz']:.3f} GHz")
    ps = wl.mc_poly_pair_scene()
    for _ in range(10):
        eng.mc_poly_pair(ps["robot"], ps["pos"], ps["theta"], ps["obstacle"], ps["std_dev"], 1234, 0, 0, 100_000_000, d)
    eng.synchronize()
    clock(eng, 0, poly=True)
    for _ in range(10):
        eng.mc_poly_pair(ps["robot"], ps["pos"], ps["theta"], ps["obstacle"], ps["std_dev"], 1234, 0, 0, 100_000_000, d)
    cp = clock(eng, 0, poly=True)
    print(f"polygon bench scene (mc_poly_pair_kernel, 10 x 1e8 samples back to back): {cp['waves']} waves, held {cp['ghz']:.3f} GHz")
    c4 = scenes(eng, 4_000_000, 120_000)
    print(f"config 4 shard (mc_scenes_advance_kernel, 4e6 data points, max_samples 120 000): {c4['waves']} waves, held {c4['ghz']:.3f} GHz")
    cd = scenes(eng, 100_000, 4_020_000)
    print(f"reference-default batch (1e5 data points, max_samples 4 020 000): {cd['waves']} waves, held {cd['ghz']:.3f} GHz")
    cq = poly_scenes(eng, 200_000)
    print(f"adaptive polygon scenes (mc_poly_scenes_advance_kernel, 2e5 scenes, max_samples 120 000): {cq['waves']} waves, held {cq['ghz']:.3f} GHz")
    if "--record" in sys.argv:
        tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "")
        path = os.path.join(ROOT, "profiles", "measured_counts.json")
        cur = json.load(open(path))
        for key, c in (("mc_pair.config3", c3), ("mc_poly_pair.bench", cp), ("mc_scenes.config4", c4), ("mc_poly_scenes.bench", cq)):
            if key in cur:
                cur[key]["held_clock_ghz"] = round(c["ghz"], 3)
                cur[key]["held_clock_source"] = ("tests/tools/mc_clock.py on the clock build (make lib-mcclock): s_memtime / s_memrealtime stamps around the "
                                                 "sample work of %d waves%s" % (c["waves"], (", " + tag) if tag else ""))
        json.dump(cur, open(path, "w"), indent=1)
        print("recorded", {k: cur[k].get("held_clock_ghz") for k in ("mc_pair.config3", "mc_poly_pair.bench", "mc_scenes.config4", "mc_poly_scenes.bench") if k in cur})
    eng.close()


if __name__ == "__main__":
    main()
