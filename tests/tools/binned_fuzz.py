#!/usr/bin/env python3
"""Developer tool: differential fuzz of the binned polygon path against the CPU oracle — random bin lists (row counts 1..16 on
either side, bin sizes around the wave / tile boundaries, exact and counted bins, padded strides), random densities, and random
padded batches through the device binning at every granularity.  Every boolean and every count must equal the oracle's.
TEST INFRASTRUCTURE (uses oracle/).   usage: binned_fuzz.py [configs] [seed]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402

spec = importlib.util.spec_from_file_location("binned_tests", os.path.join(ROOT, "tests", "test_gpu_poly_binned.py"))
bt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bt)


def one(eng, rng, it, seed, announce=None):
    """One configuration (the generator's `seed` also keys the polygons); `announce(text)` is called with its description BEFORE
    any GPU work, so that a fault names its input.  Returns (ok, what, pairs)."""
    if it % 3 == 2:  # a padded batch through the device binning
        n = int(rng.choice([1, 63, 64, 65, 1000, 4095, 4096, 4097, 20_000, 70_001]))
        rows = int(rng.integers(1, 17))
        g = int(rng.integers(1, rows + 1))
        kmin = int(rng.integers(1, rows + 1))
        if announce is not None:
            announce(f"config {it}: from_padded n={n} rows={rows} g={g} kmin={kmin}")
        vx, vy, k = wl.random_convex_polygons(n, seed=seed * 100_000 + it, kmin=kmin, kmax=rows, extent=float(rng.choice([0.5, 1.5, 6.0])), rows=rows)
        ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
        out, cnt, bins = bt.run_from_padded(eng, vx, vy, k, rows, g)
        ok = np.array_equal(out, ref) and cnt == ref_cnt
        bins.close()
        return ok, ("from_padded", it, n, rows, g), n
    nb = int(rng.integers(1, 14))
    specs = [(int(rng.integers(1, 17)), int(rng.integers(1, 17)), int(rng.choice([1, 5, 63, 64, 65, 300, 2000, 5000]))) for _ in range(nb)]
    counted = bool(rng.integers(0, 2))
    extent = float(rng.choice([0.5, 1.0, 2.0, 6.0]))
    if announce is not None:
        announce(f"config {it}: bins counted={counted} extent={extent} {specs}")
    bins, host, bufs = bt._upload_user_bins(eng, rng, specs, extent, wl, counted, int(rng.choice([0, 0, 3, 64])), seed=seed * 100_000 + it * 100)
    h = eng.poly_bins_create(bins)
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_binned(h, d_cnt)
    total, pairs, ok, what = 0, 0, True, ("bins", it, counted, extent)
    for d, (vx, vy, k), (ra, rb, n) in zip(bins, host, specs):
        ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
        if not np.array_equal(d["out"].get()[:n], ref):
            ok, what = False, ("bins", it, ra, rb, n, counted, extent)
        total += ref_cnt
        pairs += n
    if int(d_cnt.get()[0]) != total:
        ok, what = False, ("count", it)
    eng.check_async()
    h.close()
    for b in bufs + [d_cnt]:
        b.free()
    return ok, what, pairs


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    eng = pkg.Engine(0)
    pairs = 0
    verbose = bool(os.environ.get("C2D_FUZZ_VERBOSE"))   # every configuration is named BEFORE it runs: a GPU fault then names its input
    for it in range(configs):
        ok, what, n = one(eng, rng, it, seed, (lambda t: print(t, flush=True)) if verbose else None)
        assert ok, what
        pairs += n
        if it % 25 == 24:
            print(f"{it + 1} configurations, {pairs} pairs: 0 differences", flush=True)
    print(f"binned fuzz ok: {configs} configurations, {pairs} pairs, 0 differences")


if __name__ == "__main__":
    main()
