#!/usr/bin/env python3
"""Developer tool: a padded polygon batch f32[2][rows][n] is ONE counted bin of the binned entry point (ax = vx, bx = vx + rows n,
ka = k, kb = k + n, stride = n).  Times c2d_sat_poly_pairs_rows against c2d_sat_poly_pairs_binned on the same buffers for several
row layouts, and checks that the booleans are equal.  usage: onebin_bench.py [pairs] [reps]"""
import os
import sys
import time

import torch  # before libc2d.so
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import torch_random_convex_polygons  # noqa: E402

pkg = load_package()


def timed(stream, fn, reps):
    w0 = time.perf_counter()
    while time.perf_counter() - w0 < 0.15:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    bad = 0
    for rows, kmin, kmax, extent, sort in ((16, 3, 16, 8.0, False), (16, 16, 16, 8.0, False), (16, 3, 16, 1.0, False), (16, 3, 16, 8.0, True),
                                           (12, 3, 12, 8.0, False), (8, 3, 8, 8.0, False), (8, 3, 8, 1.0, False), (4, 3, 4, 8.0, False)):
        vx, vy, kk = torch_random_convex_polygons(torch, dev, n, seed=rows * 7 + kmax, kmin=kmin, kmax=kmax, extent=extent, rows=rows)
        if sort:
            order = torch.argsort(kk[0].to(torch.int64) * 32 + kk[1].to(torch.int64))
            vx, vy, kk = vx[:, :, order].contiguous(), vy[:, :, order].contiguous(), kk[:, order].contiguous()
        out_a = torch.zeros(n, dtype=torch.uint8, device=dev)
        out_b = torch.zeros(n, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        bins = eng.poly_bins_create([{"rows_a": rows, "rows_b": rows, "n": n, "ax": vx[0].data_ptr(), "ay": vy[0].data_ptr(), "bx": vx[1].data_ptr(),
                                      "by": vy[1].data_ptr(), "ka": kk[0].data_ptr(), "kb": kk[1].data_ptr(), "out": out_b.data_ptr()}])

        def padded():
            eng.sat_poly_pairs_rows(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, rows, out_a.data_ptr(), cnt.data_ptr(), stream=sh)

        def onebin():
            eng.sat_poly_pairs_binned(bins, cnt.data_ptr(), stream=sh)

        ta, tb = timed(stream, padded, reps), timed(stream, onebin, reps)
        torch.cuda.synchronize()
        eng.check_async()
        diff = int((out_a != out_b).sum().item())
        bad += diff
        rate = float(out_a.sum(dtype=torch.int64).item()) / n
        print(f"rows {rows:2d} K~U{{{kmin}..{kmax}}} extent {extent} {'sorted ' if sort else ''}(collide {rate:.3f}): rows entry point {ta:.4f} ms, "
              f"one counted bin {tb:.4f} ms ({ta / tb:.2f}x); differing booleans {diff}")
        bins.close()
        del vx, vy, kk
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
