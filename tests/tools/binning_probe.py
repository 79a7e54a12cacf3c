#!/usr/bin/env python3
"""Developer tool: c2d_poly_bins_from_padded on the config-5 workload, a few calls per granularity (run it under
`rocprofv3 --kernel-trace --stats` to see the count / scan / move kernels by themselves; wall times include hipMalloc / hipFree
of the bins' block).  usage: binning_probe.py [pairs] [granularities]"""
import os
import sys
import time

import torch  # before libc2d.so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import torch_random_convex_polygons  # noqa: E402

pkg = load_package()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    grans = [int(g) for g in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    vx, vy, kk = torch_random_convex_polygons(torch, dev, n, seed=0xC0FFEE)
    torch.cuda.synchronize()
    for g in grans:
        for rep in range(4):
            t0 = time.perf_counter()
            bins = eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, 16, g)
            t1 = time.perf_counter()
            bins.close()
            t2 = time.perf_counter()
            print(f"g={g} call {rep}: from_padded {1e3 * (t1 - t0):.2f} ms wall, destroy {1e3 * (t2 - t1):.2f} ms")


if __name__ == "__main__":
    main()
