#!/usr/bin/env python3
"""Differential fuzz of c2d_sat_poly_pairs_rows against the CPU oracle: random batch sizes, row layouts, vertex-count
ranges, scene densities, clockwise polygons, junk in the padded slots.  TEST INFRASTRUCTURE (uses oracle/).
usage: poly_fuzz.py [configs] [seed]"""
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


def one(eng, rng, idx, announce=None):
    """One configuration; `announce(text)` is called with its description BEFORE any GPU work, so that a fault names its input."""
    rows = int(rng.integers(1, 17))
    kmax = int(rng.integers(1, rows + 1))
    kmin = int(rng.integers(1, kmax + 1))
    n = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, int(rng.integers(1, 5000)), int(rng.integers(5000, 300_000))]))
    extent = float(rng.choice([0.2, 0.5, 1.0, 2.0, 4.0, 8.0]))
    pseed = int(rng.integers(1 << 30))
    if announce is not None:
        announce(f"config {idx}: sat_poly_pairs_rows n {n} rows {rows} k {kmin}..{kmax} extent {extent} polygon seed {pseed}")
    vx, vy, k = wl.random_convex_polygons(n, seed=pseed, kmin=kmin, kmax=kmax, extent=extent, rows=rows)
    # reverse the orientation of a random subset
    for p in range(2):
        flip = np.flatnonzero(rng.random(n) < 0.3)
        for kk in range(2, rows + 1):
            sel = flip[k[p][flip] == kk]
            if sel.size:
                vx[p][:kk, sel] = vx[p][:kk, sel][::-1]
                vy[p][:kk, sel] = vy[p][:kk, sel][::-1]
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    # junk in the padded slots (after the oracle ran: it must not matter, and the oracle never reads them)
    junk = np.array([np.nan, np.inf, -np.inf, 3e38, -1e-40, 0.0], np.float32)
    for p in range(2):
        mask = np.arange(rows)[:, None] >= k[p][None, :]
        m = int(mask.sum())
        if m:
            vx[p][mask] = rng.choice(junk, size=m)
            vy[p][mask] = rng.choice(junk, size=m)
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    d_out, d_cnt = eng.zeros(n + 8, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_rows(dvx, dvy, dk, n, rows, d_out, d_cnt)
    out, cnt = d_out.get(), int(d_cnt.get()[0])
    for a_ in (dvx, dvy, dk, d_out, d_cnt):
        a_.free()
    ok = np.array_equal(out[:n], ref) and cnt == ref_cnt and not out[n:].any()
    if not ok:
        bad = np.flatnonzero(out[:n] != ref)
        print(f"MISMATCH config {idx}: rows {rows} k {kmin}..{kmax} n {n} extent {extent}: {bad.size} booleans differ (first {bad[:5]}), count {cnt} vs {ref_cnt}")
    return ok, (rows, kmin, kmax, n, extent, float(ref.mean()))


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    rng = np.random.default_rng(seed)
    eng = pkg.Engine(0)
    fails = 0
    rates = []
    for i in range(configs):
        ok, info = one(eng, rng, i)
        fails += not ok
        rates.append(info[-1])
        if (i + 1) % 50 == 0:  # a sign of life for long runs
            print(f"  {i + 1} / {configs} configurations, {fails} failures so far", flush=True)
    print(f"{configs} configurations, {fails} failures; collide rate min {min(rates):.3f} median {np.median(rates):.3f} max {max(rates):.3f}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
