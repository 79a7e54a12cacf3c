#!/usr/bin/env python3
"""Differential fuzz of the pair kernels against the CPU oracle on pairs whose two shapes and whose offset each get their OWN random scale
(1e-14 .. 1e7): rectangle pairs in the pose and the vertex format, polygon pairs in the padded layout and as a binned batch.
TEST INFRASTRUCTURE (uses oracle/).   usage: pair_mixed_scale_fuzz.py [seed] [pairs]"""
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

eng = pkg.Engine(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rng = np.random.default_rng(seed)


def scales(m):
    return (10.0 ** rng.uniform(-14, 7, m)).astype(np.float32)


# ---- rectangles: sizes of A, sizes of B and the offset at independent scales; half of the pairs are pulled within reach of each other
pp = wl.random_obb_pose_planes(n, seed=seed, extent=1.0)
sa, sb, so = scales(n), scales(n), scales(n)
near = rng.random(n) < 0.5
so = np.where(near, np.maximum(sa, sb), so).astype(np.float32)
pp[2] *= sa; pp[3] *= sa; pp[7] *= sb; pp[8] *= sb
pp[0] = 0; pp[1] = 0
pp[5] *= so * 3; pp[6] *= so * 3
shift = (rng.uniform(-1, 1, (2, n)) * scales(n) * (rng.random(n) < 0.3)).astype(np.float32)   # both far from the origin, sometimes
pp[0] += shift[0]; pp[5] += shift[0]; pp[1] += shift[1]; pp[6] += shift[1]
with np.errstate(all="ignore"):
    ref_p, _ = oracle.sat_rect_pairs_pose(pp)
    planes = np.concatenate([oracle.rects_from_poses(*pp[:5]), oracle.rects_from_poses(*pp[5:])])
    ref_v, cv = oracle.sat_rect_pairs_verts(planes)
d_pp = [eng.to_device(np.ascontiguousarray(pp[k])) for k in range(10)]
d_pl = [eng.to_device(np.ascontiguousarray(planes[k])) for k in range(16)]
d_out = eng.zeros(n, np.uint8)
eng.sat_rect_pairs_pose(d_pp, n, d_out)
got_p = d_out.get()
eng.sat_rect_pairs_verts(d_pl, n, d_out)
got_v = d_out.get()
d_mask = eng.zeros((n + 63) // 64, np.uint64)
eng.sat_rect_pairs_verts_mask(d_pl, n, d_mask)
bits = np.unpackbits(d_mask.get().view(np.uint8), bitorder="little")[:n]
print(f"rectangles, {n} pairs ({cv} colliding): pose format differs on {int((got_p != ref_p).sum())}, vertex format on {int((got_v != ref_v).sum())}, "
      f"bit-mask output on {int((bits != ref_v).sum())}")
d_mask.free()
for a in d_pp + d_pl + [d_out]:
    a.free()

# ---- polygons
m = n // 4
vx, vy, k = wl.random_convex_polygons(m, seed=seed + 1, kmin=1, kmax=16, extent=1.0)
sa, sb, so = scales(m), scales(m), scales(m)
so = np.where(rng.random(m) < 0.5, np.maximum(sa, sb), so).astype(np.float32)
for p, s in ((0, sa), (1, sb)):  # centre each polygon on its own first vertex, scale it, then offset polygon B
    x0, y0 = vx[p][0].copy(), vy[p][0].copy()
    vx[p] = (vx[p] - x0) * s
    vy[p] = (vy[p] - y0) * s
off = (rng.uniform(-2, 2, (2, m)) * so).astype(np.float32)
vx[1] += off[0]; vy[1] += off[1]
vx, vy = np.ascontiguousarray(vx, np.float32), np.ascontiguousarray(vy, np.float32)
with np.errstate(all="ignore"):
    ref, c = oracle.sat_poly_pairs(vx, vy, k)
dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
d_out = eng.zeros(m, np.uint8)
eng.sat_poly_pairs(dvx, dvy, dk, m, d_out)
got = d_out.get()
bins = eng.poly_bins_from_padded(dvx, dvy, dk, m, 16, 1)
eng.sat_poly_pairs_binned(bins, None)
d_bo = eng.zeros(m, np.uint8)
bins.results(d_bo)
got_b = d_bo.get()
bins.close()
print(f"polygons, {m} pairs ({c} colliding): padded layout differs on {int((got != ref).sum())}, binned batch on {int((got_b != ref).sum())}")
for a in (dvx, dvy, dk, d_out, d_bo):
    a.free()

# ---- triangles and quadrilaterals in the 4-row layout (its own kernel)
vx4, vy4, k4 = wl.random_convex_polygons(m, seed=seed + 2, kmin=1, kmax=4, extent=1.0, rows=4)
sa, sb, so = scales(m), scales(m), scales(m)
so = np.where(rng.random(m) < 0.5, np.maximum(sa, sb), so).astype(np.float32)
for p, s_ in ((0, sa), (1, sb)):
    x0, y0 = vx4[p][0].copy(), vy4[p][0].copy()
    vx4[p] = (vx4[p] - x0) * s_
    vy4[p] = (vy4[p] - y0) * s_
off = (rng.uniform(-2, 2, (2, m)) * so).astype(np.float32)
vx4[1] += off[0]; vy4[1] += off[1]
vx4, vy4 = np.ascontiguousarray(vx4, np.float32), np.ascontiguousarray(vy4, np.float32)
with np.errstate(all="ignore"):
    pad = lambda a: np.concatenate([a, np.zeros((2, 12, m), np.float32)], axis=1)
    ref4, c4 = oracle.sat_poly_pairs(pad(vx4), pad(vy4), k4)
dvx, dvy, dk = eng.to_device(vx4), eng.to_device(vy4), eng.to_device(k4)
d_out = eng.zeros(m, np.uint8)
eng.sat_poly_pairs_rows(dvx, dvy, dk, m, 4, d_out)
print(f"4-row polygons, {m} pairs ({c4} colliding): differ on {int((d_out.get() != ref4).sum())}")
