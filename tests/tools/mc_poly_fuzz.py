#!/usr/bin/env python3
"""Differential fuzz of c2d_mc_poly_scenes / c2d_mc_poly_pair against the CPU oracle: random robot polygons and obstacle tables
(1 to 16 vertices, clockwise and counter-clockwise, points and segments, slivers, whole scenes at other scales), standard deviations
with and without shape noise and with zeros, scene counts, accuracy bins, max_samples and sampling schedules; per-scene hit counts,
sample counts and output rows must be equal bit for bit.  TEST INFRASTRUCTURE (uses oracle/).   usage: mc_poly_fuzz.py [configs] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402


def one(eng, rng, idx):
    nrng = np.random.Generator(np.random.Philox(int(rng.integers(1 << 40))))
    ntab = int(rng.integers(1, 120))
    kmin = int(rng.choice([1, 2, 3, 3, 3]))
    kmax = int(rng.integers(max(kmin, 3), 17))
    poses, sds = wl.random_poly_tables(ntab, ntab, seed=int(rng.integers(1 << 30)), kmin=kmin, kmax=kmax, shape_variance=bool(rng.integers(2)))
    if rng.random() < 0.25:  # some zero standard deviations (deterministic coordinates)
        sds = sds.copy()
        sds["x"][: ntab // 2] = 0
        sds["theta"][: ntab // 3] = 0
        sds["y"][ntab // 2:] = 0
    ka = int(rng.integers(1, 17))
    robot = wl.convex_polygon(ka, nrng, float(rng.uniform(0.3, 3)), float(rng.uniform(0.3, 3)), float(rng.uniform(0, 6.28)), clockwise=bool(rng.integers(2)))
    if rng.random() < 0.15:  # a sliver robot
        robot = (robot[0], (robot[1] * np.float32(1e-4)).astype(np.float32))
    scale = np.float32(1.0)
    if rng.random() < 0.3:  # the whole scene at another scale
        # (one scene in five of these far out: denormal products below, the edge of the fast paths' domain above)
        scale = np.float32(10.0 ** int(rng.integers(-6, 7) if rng.random() < 0.8 else rng.choice([-30, -26, -23, -22, -21, -19, -16, -14, 7, 8, 11])))
        poses, sds = poses.copy(), sds.copy()
        poses["obstacle"]["x"] *= scale
        poses["obstacle"]["y"] *= scale
        for f in ("x", "y"):
            sds[f] *= scale
        robot = ((robot[0] * scale).astype(np.float32), (robot[1] * scale).astype(np.float32))
    ns = int(rng.choice([1, 2, 63, 65, int(rng.integers(1, 300)), int(rng.integers(300, 1500))]))
    rr = float(np.sqrt(robot[0].astype(np.float64) ** 2 + robot[1].astype(np.float64) ** 2).max())
    scenes = wl.random_poly_scenes(ns, poses, sds, rr, seed=int(rng.integers(1 << 30)), spread=float(rng.choice([0.5, 2.0, 4.0, 8.0])))
    if rng.random() < 0.3:  # far scenes too: the radius-word path
        scenes["x"] *= np.float32(rng.uniform(1.5, 4))
        scenes["y"] *= np.float32(rng.uniform(1.5, 4))
    seed, base = int(rng.integers(1 << 40)), int(rng.integers(1 << 33))
    schedule = [(0, 0, 0), (10000, 10000, 0), (64, 1000, 640), (100, 7777, 1000), (1000, 33333, 5000), (1500, 100000, 3000)][int(rng.integers(6))]
    max_samples = int(rng.choice([1000, 3000, 20000, 50000, 120000]))
    if ns * max_samples > 30_000_000:  # (the oracle walks a polygon sample in ~1 us per core: keep a configuration within seconds)
        max_samples = max(1000, 30_000_000 // ns // 1000 * 1000)
    nb = int(rng.integers(2, 6))
    bins = np.concatenate([[0.0], np.sort(rng.uniform(0.001, 0.9, nb - 2)), [1.0]]).astype(np.float32)
    acc = np.sort(rng.uniform(5e-4, 5e-2, nb - 1)).astype(np.float32)
    h_ref, u_ref, rows_ref, tot_ref = oracle.mc_poly_scenes(robot, poses, sds, scenes, bins, acc, max_samples, seed, base, schedule=schedule)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u, d_r = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32), eng.empty(ns, pkg.ROW_DT)
    tot, _ = eng.mc_poly_scenes(robot, d_p, ntab, d_s, ntab, d_sc, ns, bins, acc, max_samples, seed, base, d_h, d_u, d_r, schedule=schedule)
    ok = (np.array_equal(d_h.get(), h_ref) and np.array_equal(d_u.get(), u_ref) and np.array_equal(d_r.get().view(np.uint32), rows_ref.view(np.uint32))
          and tot == tot_ref)
    # one scene of the batch through the sample-parallel entry point, an odd sample range
    j = int(rng.integers(ns))
    pi, vi = int(scenes["pose_idx"][j]), int(scenes["var_idx"][j])
    begin, count = int(rng.integers(1 << 34)), int(rng.integers(1, 150_000))
    kb = int(poses["obstacle"]["k"][pi])
    obstacle = (poses["obstacle"]["x"][pi][:kb].copy(), poses["obstacle"]["y"][pi][:kb].copy())
    sd = tuple(float(v) for v in sds[vi])
    pos, theta = (float(scenes["x"][j]), float(scenes["y"][j])), float(poses["theta"][pi])
    d_hits = eng.zeros(1, np.uint64)
    eng.mc_poly_pair(robot, pos, theta, obstacle, sd, seed, base + j, begin, count, d_hits)
    ok = ok and int(d_hits.get()[0]) == oracle.mc_poly_pair(robot, pos, theta, obstacle, sd, seed, base + j, begin, count)
    for a in (d_p, d_s, d_sc, d_h, d_u, d_r, d_hits):
        a.free()
    if not ok:
        print(f"MISMATCH config {idx}: scenes {ns} tables {ntab} ka {ka} k {kmin}..{kmax} scale {scale} schedule {schedule} max_samples {max_samples}")
    return ok, tot_ref + count


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    eng = pkg.Engine(0)
    fails, total = 0, 0
    for i in range(configs):
        ok, n = one(eng, rng, i)
        fails += not ok
        total += n
        if (i + 1) % 25 == 0:  # a sign of life for long runs
            print(f"  {i + 1} / {configs} configurations, {fails} failures so far", flush=True)
    print(f"{configs} configurations, {fails} failures, {total:.3e} samples checked")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
