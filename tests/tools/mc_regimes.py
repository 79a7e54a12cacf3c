#!/usr/bin/env python3
"""Developer tool: throughput of c2d_mc_pair on single scenes that each exercise ONE regime of the Monte-Carlo kernels — the
near path with (nearly) every sample ruled out by its centre, with some, with none; the far path with few and with many radius
candidates; the config-3 scene — so that a regime's cost can be set against its arithmetic ("Near scenes").
usage: mc_regimes.py [lib.so ...]   (several libraries = an A/B of builds in one process)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()

ROBOT = (4.07, 1.74)
SCENES = [  # name, robot position, obstacle (w, h, theta), standard deviations (x, y, theta, w, h)
    ("config-3 scene (p = 0.55, 65 % evaluated)", (3.0, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),
    ("near, nearly every sample ruled out by its centre (sigma 30)", (3.0, 1.0), (2.0, 1.0, 0.6), (30.0, 30.0, 0.2, 0.0, 0.0)),
    ("near, four in five ruled out (sigma 3)", (3.0, 1.0), (2.0, 1.0, 0.6), (3.0, 3.0, 0.2, 0.0, 0.0)),
    ("near, p ~ 0.1 (sigma 1.2 at a distance)", (4.2, 1.0), (2.0, 1.0, 0.6), (1.2, 1.2, 0.2, 0.0, 0.0)),
    ("near, nothing ruled out, every sample collides", (1.0, 0.3), (2.0, 1.0, 0.6), (0.01, 0.01, 0.2, 0.0, 0.0)),
    ("far, one radius candidate in 10^4", (9.0, 1.0), (2.0, 1.0, 0.6), (0.8, 0.8, 0.2, 0.0, 0.0)),
    ("far, three radius candidates in ten", (5.0, 1.0), (2.0, 1.0, 0.6), (1.2, 1.2, 0.2, 0.0, 0.0)),
]


def main():
    libs = sys.argv[1:] or [pkg.library_path()]
    n = 2_000_000_000
    for lib in libs:
        eng = pkg.Engine(0, lib_path=os.path.abspath(lib))
        st = eng.stream_create()
        print(os.path.basename(lib), flush=True)
        for name, pos, pose, sd in SCENES:
            d = eng.zeros(1, np.uint64)
            best = 1e9
            for _ in range(3):
                eng.memset(d.ptr, 0, 8)
                eng.synchronize()
                t0 = time.perf_counter()
                eng.mc_pair(ROBOT[0], ROBOT[1], pos, pose, sd, 1234, 0, 0, n, d, stream=st)
                eng.synchronize(st)
                best = min(best, time.perf_counter() - t0)
            hits = int(d.get()[0])
            print(f"  {name}: {n / best / 1e9:.1f}e9 samples/s, p = {hits / n:.5f} (hits {hits})", flush=True)
            d.free()
        eng.close()


if __name__ == "__main__":
    main()
