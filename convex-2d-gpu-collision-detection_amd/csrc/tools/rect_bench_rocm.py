#!/usr/bin/env python3
"""Developer tool: rect_bench.py's measurement in a process that holds ROCm's OWN HIP runtime (no torch): c2d_sat_rect_pairs_verts on
the config-2 workload with and without the colliding count, HIP events (through ctypes) around 200 back-to-back calls after a
pre-warm.  With the count the workspace guard runs — on this runtime with one hipStreamGetId per call (csrc/c2d_internal.hpp) —
without it the guard is not involved; the same pair of numbers from a PyTorch process (HIP 7.0, no stream ids) is rect_bench.py's.
usage: rect_bench_rocm.py [lib.so ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")


def main():
    libs = sys.argv[1:] or [pkg.library_path()]
    engs = [pkg.Engine(0, lib_path=os.path.abspath(p)) for p in libs]
    hip = C.CDLL("libamdhip64.so.7")
    print("hipStreamGetId", "present" if hasattr(hip, "hipStreamGetId") else "absent", "in this process' HIP runtime")
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    eng = engs[0]
    n = 10_000_000
    poses = wl.random_obb_pose_planes(n, seed=0x5A7)
    d_pose = eng.to_device(poses)
    planes = eng.empty((16, n), np.float32)
    for r in range(2):
        eng.rects_from_poses(*[d_pose.row(5 * r + k) for k in range(5)], n, [planes.row(8 * r + k) for k in range(8)])
    out, cnt = eng.empty(n, np.uint8), eng.zeros(1, np.uint64)
    s = eng.stream_create()
    eng.synchronize()
    ptrs = [planes.row(k) for k in range(16)]
    e0, e1 = C.c_void_p(), C.c_void_p()
    assert hip.hipEventCreate(C.byref(e0)) == 0 and hip.hipEventCreate(C.byref(e1)) == 0

    def timed(fn, reps=200):
        w0 = time.perf_counter()
        while time.perf_counter() - w0 < 0.15:
            for _ in range(20):
                fn()
            eng.synchronize(s)
        assert hip.hipEventRecord(e0, C.c_void_p(s)) == 0
        for _ in range(reps):
            fn()
        assert hip.hipEventRecord(e1, C.c_void_p(s)) == 0
        assert hip.hipEventSynchronize(e1) == 0
        ms = C.c_float()
        assert hip.hipEventElapsedTime(C.byref(ms), e0, e1) == 0
        return ms.value / reps * 1e3

    for rep in range(3):
        for p, e in zip(libs, engs):
            a = timed(lambda: e.sat_rect_pairs_verts(ptrs, n, out, cnt, stream=s))
            b = timed(lambda: e.sat_rect_pairs_verts(ptrs, n, out, None, stream=s))
            print(f"{os.path.basename(p):28s} with count {a:7.2f} us ({65 * n / a / 1e3:5.0f} GB/s)   without {b:7.2f} us ({65 * n / b / 1e3:5.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
