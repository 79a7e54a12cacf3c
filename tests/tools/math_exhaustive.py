#!/usr/bin/env python3
"""Developer tool: the device's canonical math against the CPU oracle over WHOLE input domains, bit for bit —
  sin/cos of 2 pi y / 2^32       every one of the 2^32 angle words a Philox block can hold;
  log(u)                          every float in [2^-33, 1]: a superset of u = (x + 1/2) 2^-32 over all radius words x;
  sqrt(v)                         every float in [2^-24, 64]: a superset of -2 log(u) over that range;
  sin/cos(x) of a float angle     every float with |x| <= 128 (subnormals included) and every 7th float up to 2^31;
  Box-Muller                      both normals of (word, hash(word)) for every 32-bit word.
With these equal, a Monte-Carlo sample's Gaussian draws are the oracle's for every Philox output, and what remains of a hit count
is float geometry.  TEST INFRASTRUCTURE (uses oracle/).   usage: math_exhaustive.py   (about two minutes on the GPU box's 16 cores)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from oracle import cpu as oracle  # noqa: E402

CHUNK = 1 << 26


def sweep(eng, fn, name, ranges, step=1):
    """ranges: [(first_bits, last_bits)] inclusive, as uint32 bit patterns"""
    d_in, d0, d1 = eng.empty(CHUNK, np.uint32), eng.empty(CHUNK, np.float32), eng.empty(CHUNK, np.float32)
    total, bad, t0 = 0, 0, time.perf_counter()
    for lo, hi in ranges:
        pos = lo
        while pos <= hi:
            n = min(CHUNK, (hi - pos) // step + 1)
            bits = (np.arange(n, dtype=np.uint64) * step + pos).astype(np.uint32)
            eng.lib.c2d_memcpy_h2d(eng.h, d_in.ptr, bits.ctypes.data, bits.nbytes, None)
            eng.math_eval(fn, d_in, n, d0, d1)
            g0, g1 = d0.get()[:n], d1.get()[:n]
            r0, r1 = oracle.math_eval(fn, bits)
            same = (g0.view(np.uint32) == r0.view(np.uint32)) & (g1.view(np.uint32) == r1.view(np.uint32))
            bad += int(n - same.sum())
            total += n
            pos += n * step
    print(f"{name}: {total} inputs, {bad} differing results ({time.perf_counter() - t0:.1f} s)", flush=True)
    for a in (d_in, d0, d1):
        a.free()
    return bad


def fbits(x):
    return int(np.float32(x).view(np.uint32))


def main():
    eng = pkg.Engine(0)
    oracle.set_num_threads(oracle.usable_cores())
    bad = 0
    bad += sweep(eng, eng.MATH_SINCOS_U32, "sin/cos of 2 pi y / 2^32, every 32-bit word y", [(0, 0xFFFFFFFF)])
    bad += sweep(eng, eng.MATH_BOX_MULLER, "Box-Muller normals of (word, hash(word)), every 32-bit word", [(0, 0xFFFFFFFF)])
    bad += sweep(eng, eng.MATH_LOG, "log(u), every float in [2^-33, 1]", [(fbits(2.0 ** -33), fbits(1.0))])
    bad += sweep(eng, eng.MATH_SQRT, "sqrt(v), every float in [2^-24, 64]", [(fbits(2.0 ** -24), fbits(64.0))])
    bad += sweep(eng, eng.MATH_SINCOS, "sin/cos(x), every float with |x| <= 128", [(0, fbits(128.0)), (0x80000000, 0x80000000 + fbits(128.0))])
    bad += sweep(eng, eng.MATH_SINCOS, "sin/cos(x), every 7th float with 128 < |x| <= 2^31",
                 [(fbits(128.0) + 1, fbits(2.0 ** 31)), (0x80000000 + fbits(128.0) + 1, 0x80000000 + fbits(2.0 ** 31))], step=7)
    print("math exhaustive:", "ok, 0 differences" if not bad else f"{bad} DIFFERENCES")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
