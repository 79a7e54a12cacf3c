#!/usr/bin/env python3
"""Differential fuzz of c2d_mc_scenes / c2d_mc_pair against the CPU oracle: random tables (with and without shape variance),
scene counts, robot sizes, spreads, accuracy bins, max_samples and sampling schedules; per-scene hit counts, sample counts
and output rows must be equal bit for bit.  TEST INFRASTRUCTURE (uses oracle/).   usage: mc_fuzz.py [configs] [seed]"""
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
    ntab = int(rng.integers(1, 200))
    tp, ts, _ = wl.random_tables(ntab, ntab, seed=int(rng.integers(1 << 30)), shape_variance=bool(rng.integers(2)))
    if rng.random() < 0.2:  # some zero standard deviations (deterministic scenes)
        ts = ts.copy()
        ts["x"][: ntab // 2] = 0
        ts["theta"][: ntab // 3] = 0
    ns = int(rng.choice([1, 2, 63, 65, int(rng.integers(1, 400)), int(rng.integers(400, 2500))]))
    rw, rh = float(rng.uniform(0.5, 5)), float(rng.uniform(0.5, 3))
    spread = float(rng.choice([0.5, 2.0, 4.0, 8.0]))
    if rng.random() < 0.3:  # the whole scene at another scale, or the obstacles as slivers: the closed-form evaluation's margins scale with both
        # (one scene in five of these far out: denormal products below, the edge of the fast paths' domain above)
        k = np.float32(10.0 ** int(rng.integers(-9, 10) if rng.random() < 0.8 else rng.choice([-30, -26, -23, -22, -21, -19, -16, -14, 12, 14])))
        tp, ts = tp.copy(), ts.copy()
        sliver = np.float32(1e-4) if rng.random() < 0.3 else np.float32(1.0)
        tp["width"] *= k
        tp["height"] *= k * sliver
        for f in ("x", "y", "width", "height"):
            ts[f] *= k
        rw, rh, spread = float(np.float32(rw) * k), float(np.float32(rh) * k), float(np.float32(spread) * k)
    seed, base = int(rng.integers(1 << 40)), int(rng.integers(1 << 33))
    schedule = [(0, 0, 0), (10000, 10000, 0), (64, 1000, 640), (100, 7777, 1000), (1000, 33333, 5000), (1500, 100000, 3000)][int(rng.integers(6))]
    max_samples = int(rng.choice([1000, 3000, 20000, 50000, 150000]))
    nb = int(rng.integers(2, 6))
    bins = np.concatenate([[0.0], np.sort(rng.uniform(0.001, 0.9, nb - 2)), [1.0]]).astype(np.float32)
    acc = np.sort(rng.uniform(5e-4, 5e-2, nb - 1)).astype(np.float32)
    if announce is not None:
        announce(f"config {idx}: scenes {ns} tables {ntab} robot {rw!r} x {rh!r} spread {spread!r} schedule {schedule} max_samples {max_samples} "
                 f"bins {bins.tolist()} accuracy {acc.tolist()} seed {seed} base {base}")
    scenes = oracle.sample_scenes(tp, ts, rw, rh, spread, seed, base, ns)
    h_ref, u_ref, rows_ref, tot_ref = oracle.mc_scenes(tp, ts, scenes, rw, rh, bins, acc, max_samples, seed + 1, base, schedule=schedule)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, ntab, d_s, ntab, rw, rh, spread, seed, base, ns, d_sc)
    d_h, d_u, d_r = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32), eng.empty(ns, pkg.ROW_DT)
    tot, _ = eng.mc_scenes(d_p, ntab, d_s, ntab, d_sc, ns, rw, rh, bins, acc, max_samples, seed + 1, base, d_h, d_u, d_r, schedule=schedule)
    ok = (np.array_equal(d_sc.get().view(np.uint32), scenes.view(np.uint32)) and np.array_equal(d_h.get(), h_ref) and np.array_equal(d_u.get(), u_ref)
          and np.array_equal(d_r.get().view(np.uint32), rows_ref.view(np.uint32)) and tot == tot_ref)
    # one scene of the batch through the sample-parallel entry point, an odd sample range
    j = int(rng.integers(ns))
    pi, vi = int(scenes["pose_idx"][j]), int(scenes["var_idx"][j])
    begin, count = int(rng.integers(1 << 34)), int(rng.integers(1, 200_000))
    pose, sd = tuple(float(v) for v in tp[pi]), tuple(float(v) for v in ts[vi])
    pos = (float(scenes["x"][j]), float(scenes["y"][j]))
    if announce is not None:
        announce(f"config {idx}: mc_pair scene {j} of it, samples [{begin}, {begin + count})")
    d_hits = eng.zeros(1, np.uint64)
    eng.mc_pair(rw, rh, pos, pose, sd, seed, base + j, begin, count, d_hits)
    ok = ok and int(d_hits.get()[0]) == oracle.mc_pair(rw, rh, pos, pose, sd, seed, base + j, begin, count)
    for a in (d_p, d_s, d_sc, d_h, d_u, d_r, d_hits):
        a.free()
    if not ok:
        print(f"MISMATCH config {idx}: scenes {ns} tables {ntab} schedule {schedule} max_samples {max_samples} bins {bins.tolist()}")
    return ok, (ns, schedule, max_samples, tot_ref)


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    eng = pkg.Engine(0)
    fails, total = 0, 0
    for i in range(configs):
        ok, info = one(eng, rng, i)
        fails += not ok
        total += info[-1]
        if (i + 1) % 50 == 0:  # a sign of life for long runs
            print(f"  {i + 1} / {configs} configurations, {fails} failures so far", flush=True)
    print(f"{configs} configurations, {fails} failures, {total:.3e} samples checked")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
