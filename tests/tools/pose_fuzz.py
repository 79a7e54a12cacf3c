#!/usr/bin/env python3
"""Differential fuzz of c2d_sat_rect_pairs_pose against the CPU oracle: pose pairs over many scales (1e-14 .. 1e14), distances
from the origin, extent ranges down to slivers, angle sets (random, tiny, near multiples of pi/2, huge), touching pairs
(workloads.touching_pose_pairs) and non-finite components; aligned and unaligned planes.  Every boolean and every count must equal
the oracle's.  The kernel decides most pairs from a closed-form gap and the rest by the vertex arithmetic (c2d_sat.hip
pose_pair_closed_form): this fuzz is aimed at the seam.  TEST INFRASTRUCTURE (uses oracle/).   usage: pose_fuzz.py [configs] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import importlib  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402


def make(rng, n):
    kind = int(rng.integers(0, 5))
    scale = float(10.0 ** rng.integers(-14, 15)) if rng.random() < 0.5 else 1.0
    offset = float(rng.choice([0.0, 0.0, 3.0, 100.0, 1e4, 1e6]))
    if kind == 0:
        return wl.touching_pose_pairs(n, seed=int(rng.integers(1 << 30)), scale=scale, offset=offset), ("touching", scale, offset)
    ext = float(rng.choice([0.5, 2.0, 8.0, 50.0]))
    lo, hi = [(0.1, 5.0), (1e-6, 3.0), (1.0, 1.0001), (0.0, 2.0), (1e-3, 1e-2)][int(rng.integers(5))]
    p = np.empty((10, n), np.float64)
    for r in range(2):
        p[5 * r + 0] = rng.uniform(-ext, ext, n) + offset
        p[5 * r + 1] = rng.uniform(-ext, ext, n) + offset
        p[5 * r + 2] = rng.uniform(lo, hi, n) * rng.choice([1.0, 1.0, -1.0], n)   # negative extents are legal inputs too
        p[5 * r + 3] = rng.uniform(lo, hi, n)
        a = int(rng.integers(4))
        p[5 * r + 4] = (rng.uniform(-7, 7, n) if a == 0 else rng.normal(0, 1e-4, n) + rng.choice([0, np.pi / 2, np.pi, -np.pi / 2], n) if a == 1
                        else rng.uniform(-1e5, 1e5, n) if a == 2 else rng.choice([0.0, 0.5, 1.0], n))
    p[[0, 1, 2, 3, 5, 6, 7, 8]] *= scale
    p = p.astype(np.float32)
    if kind == 4:
        p = wl.inject_non_finite(p, seed=int(rng.integers(1 << 30)), frac=0.2)
    return p, (["random", "random", "random", "random", "non-finite"][kind], scale, offset, ext, lo, hi)


def one(eng, rng, idx, announce=None):
    """One configuration; `announce(text)` is called with its description BEFORE any GPU work, so that a fault names its input."""
    n = int(rng.choice([1, 255, 256, 257, 100_003, 400_000]))
    poses, what = make(rng, n)
    off = int(rng.integers(0, 2))
    if announce is not None:
        announce(f"config {idx}: sat_rect_pairs_pose n {n} plane offset {off} floats, {what}")
    ref, ref_cnt = oracle.sat_rect_pairs_pose(poses)
    host = np.zeros((10, n + 4), np.float32)
    host[:, off:off + n] = poses
    d = eng.to_device(host)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_pose([d.row(k) + 4 * off for k in range(10)], n, d_out, d_cnt)
    bad = int((d_out.get() != ref).sum())
    ok = not bad and int(d_cnt.get()[0]) == ref_cnt
    if not ok:
        print(f"MISMATCH config {idx}: {what} n {n} offset {off}: {bad} booleans differ", flush=True)
    for a in (d, d_out, d_cnt):
        a.free()
    return ok, (what[0], n, off)


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    eng = pkg.Engine(0)
    fails, pairs = 0, 0
    for i in range(configs):
        ok, info = one(eng, rng, i)
        fails += not ok
        pairs += info[1]
        if (i + 1) % 50 == 0:
            print(f"  {i + 1} / {configs} configurations, {pairs} pairs, {fails} failures so far", flush=True)
    print(f"pose fuzz: {configs} configurations, {pairs} pairs, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
