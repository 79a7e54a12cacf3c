#!/usr/bin/env python3
"""Developer tool: c2d_sat_rect_pairs_verts on the config-2 workload, with and without the colliding count, for one or several
builds of libc2d.so in ONE process (interleaved, repeated): HIP events around 200 back-to-back calls after a pre-warm.
usage: rect_bench.py [lib.so ...]"""
import os
import sys
import time

import torch  # before libc2d.so
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()


def main():
    libs = sys.argv[1:] or [pkg.library_path()]
    dev = torch.device("cuda", 0)
    n = 10_000_000
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5A7)
    pose = torch.empty((10, n), dtype=torch.float32, device=dev)
    for r in range(2):
        pose[5 * r + 0].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 1].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 2].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 3].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 4].uniform_(0.0, 2.0 * np.pi, generator=gen)
    planes = torch.empty((16, n), dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    row = lambda t, k: t.data_ptr() + k * t.stride(0) * t.element_size()  # noqa: E731
    engs = [pkg.Engine(0, lib_path=os.path.abspath(p)) for p in libs]
    torch.cuda.synchronize()
    for r in range(2):
        engs[0].rects_from_poses(*[row(pose, 5 * r + k) for k in range(5)], n, [row(planes, 8 * r + k) for k in range(8)], stream=sh)
    torch.cuda.synchronize()
    ptrs = [row(planes, k) for k in range(16)]

    def timed(fn, reps=200):
        w0 = time.perf_counter()
        while time.perf_counter() - w0 < 0.15:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for rep in range(3):
        for p, e in zip(libs, engs):
            a = timed(lambda: e.sat_rect_pairs_verts(ptrs, n, out.data_ptr(), cnt.data_ptr(), stream=sh))
            b = timed(lambda: e.sat_rect_pairs_verts(ptrs, n, out.data_ptr(), None, stream=sh))
            print(f"{os.path.basename(p):28s} with count {a:7.2f} us ({65 * n / a / 1e3:5.0f} GB/s)   without {b:7.2f} us ({65 * n / b / 1e3:5.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
