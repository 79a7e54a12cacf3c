#!/usr/bin/env python3
"""Developer tool: rectangle-pair batches that start and end in HOST memory — c2d_sat_rect_pairs_verts_host / _pose_host from
page-locked and from pageable buffers, against the same steps issued by hand through the device entry points — and the
host-to-device link itself (one large page-locked copy), so that the rates can be read as fractions of the link.
usage: host_batch_bench.py [pairs] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()


def best(fn, reps):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t0)
    return b


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    eng = pkg.Engine(0)
    rng = np.random.Generator(np.random.Philox(1))
    res = {}
    # the link: one page-locked copy of 640 MB each way
    big = eng.host_empty(16 * n, np.float32)
    big[:] = 1.0
    d_big = eng.empty(16 * n, np.float32)
    import ctypes as C

    def h2d():
        eng._check(eng.lib.c2d_memcpy_h2d(eng.h, C.c_void_p(d_big.ptr), C.c_void_p(big.ctypes.data), big.nbytes, None), "h2d")
        eng.synchronize()

    def d2h():
        eng._check(eng.lib.c2d_memcpy_d2h(eng.h, C.c_void_p(big.ctypes.data), C.c_void_p(d_big.ptr), big.nbytes, None), "d2h")
        eng.synchronize()

    res["h2d_GBs"] = big.nbytes / best(h2d, reps) / 1e9
    res["d2h_GBs"] = big.nbytes / best(d2h, reps) / 1e9
    print(f"link, one page-locked copy of {big.nbytes / 1e6:.0f} MB: host to device {res['h2d_GBs']:.1f} GB/s, device to host {res['d2h_GBs']:.1f} GB/s", flush=True)
    for fmt, planes, bpp in (("pose", 10, 40), ("verts", 16, 64)):
        src = rng.uniform(0.1, 5.0, (planes, n)).astype(np.float32)
        pinned = [eng.host_empty(n, np.float32) for _ in range(planes)]
        for k in range(planes):
            pinned[k][:] = src[k]
        out_p = eng.host_empty(n, np.uint8)
        out = np.zeros(n, np.uint8)
        t = best(lambda: eng.sat_rect_pairs_host(pinned, out_p, fmt), reps)
        res[fmt + "_pinned"] = n / t
        print(f"{fmt}: page-locked buffers, the host entry point: {t * 1e3:.2f} ms = {n / t / 1e9:.3f}e9 pairs/s = {bpp * n / t / 1e9:.1f} GB/s up = "
              f"{bpp * n / t / 1e9 / res['h2d_GBs']:.2f} of the link", flush=True)
        t = best(lambda: eng.sat_rect_pairs_host([src[k] for k in range(planes)], out, fmt), reps)
        res[fmt + "_pageable"] = n / t
        print(f"{fmt}: pageable buffers, the host entry point: {t * 1e3:.2f} ms = {n / t / 1e9:.3f}e9 pairs/s", flush=True)
        # the naive round trip: one stream, upload every plane, one kernel, download (page-locked buffers)
        d_in = eng.empty((planes, n), np.float32)
        d_out = eng.empty(n, np.uint8)

        def naive():
            for k in range(planes):
                eng._check(eng.lib.c2d_memcpy_h2d(eng.h, C.c_void_p(d_in.row(k)), C.c_void_p(pinned[k].ctypes.data), 4 * n, None), "h2d")
            if fmt == "pose":
                eng.sat_rect_pairs_pose([d_in.row(k) for k in range(planes)], n, d_out)
            else:
                eng.sat_rect_pairs_verts([d_in.row(k) for k in range(planes)], n, d_out)
            eng._check(eng.lib.c2d_memcpy_d2h(eng.h, C.c_void_p(out_p.ctypes.data), C.c_void_p(d_out.ptr), n, None), "d2h")
            eng.synchronize()

        t = best(naive, reps)
        print(f"{fmt}: page-locked buffers, upload - kernel - download by hand through the device entry points: {t * 1e3:.2f} ms = {n / t / 1e9:.3f}e9 pairs/s", flush=True)
        for a in pinned + [out_p]:
            eng.host_free(a)
        d_in.free()
        d_out.free()
    eng.close()


if __name__ == "__main__":
    main()
