"""sat.py — numpy restatement of the reference's rectangle SAT test.

TEST INFRASTRUCTURE ONLY (the parity checker; never imported by the product
path).  The reference README (README.md:3) names a ``SAT.py`` that is not in
the reference snapshot; this file plays that role: an
independent float32 restatement of ``create_rect`` (utils.cu:119-130) and
``convex_collide`` (utils.cu:159-184), written against the reference source,
not against oracle/c2d_oracle.c, so that the two oracles check each other
bit for bit.

PARITY UNPINNED: the reference holds no golden vectors for this path; see the
header of oracle/c2d_oracle.c.

All arithmetic is numpy float32: every ``*``, ``+``, ``-`` rounds to binary32
and nothing is fused, which is the canonical arithmetic of DESIGN.md.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
KMAX = 16


def create_rect(w, h):
    """utils.cu:119-130 — (n,) widths/heights -> (n, 8) flat x0,y0,..,x3,y3."""
    w = np.asarray(w, dtype=F32)
    h = np.asarray(h, dtype=F32)
    two = F32(2)
    r = np.empty(w.shape + (8,), dtype=F32)
    r[..., 0] = -w / two
    r[..., 1] = -h / two
    r[..., 2] = w / two
    r[..., 3] = -h / two
    r[..., 4] = w / two
    r[..., 5] = h / two
    r[..., 6] = -w / two
    r[..., 7] = h / two
    return r


def rot_trans_given_cs(r, dx, dy, c, s):
    """utils.cu:136-141 with cos/sin supplied by the caller (the canonical
    sin/cos lives in the C oracle; numpy has no float32 fma)."""
    r = np.array(r, dtype=F32, copy=True)
    dx = np.asarray(dx, dtype=F32)
    dy = np.asarray(dy, dtype=F32)
    c = np.asarray(c, dtype=F32)
    s = np.asarray(s, dtype=F32)
    for i in range(4):
        x = r[..., 2 * i].copy()
        y = r[..., 2 * i + 1].copy()
        r[..., 2 * i] = (c * x - s * y) + dx
        r[..., 2 * i + 1] = (s * x + c * y) + dy
    return r


def minmax_element(p):
    """thrust::minmax_element over the last axis (utils.cu:176-177): comparison based — both extremes start at
    element 0 and element k replaces one only when ``<`` says so, so a NaN at k > 0 is skipped and a NaN at
    k = 0 stays (numpy's own min/max would propagate every NaN)."""
    lo = p[..., 0].copy()
    hi = p[..., 0].copy()
    with np.errstate(invalid="ignore"):
        for k in range(1, p.shape[-1]):
            e = p[..., k]
            lo = np.where(e < lo, e, lo)
            hi = np.where(hi < e, e, hi)
    return lo, hi


def convex_collide(r1, r2):
    """utils.cu:159-184 for (n, 8) float32 arrays -> (n,) uint8.

    Axis = the edge vector itself (utils.cu:170-171), all eight axes always
    evaluated, separation test strict ``<`` (utils.cu:178)."""
    r1 = np.asarray(r1, dtype=F32)
    r2 = np.asarray(r2, dtype=F32)
    assert r1.shape == r2.shape and r1.shape[-1] == 8
    collide = np.ones(r1.shape[:-1], dtype=bool)
    for r in (r1, r2):
        for i in range(4):
            with np.errstate(invalid="ignore", over="ignore"):
                n0 = r[..., (i + 1) * 2 % 8] - r[..., i * 2]
                n1 = r[..., ((i + 1) * 2 + 1) % 8] - r[..., i * 2 + 1]
                p1 = np.stack([n0 * r1[..., k * 2] + n1 * r1[..., k * 2 + 1] for k in range(4)], axis=-1)
                p2 = np.stack([n0 * r2[..., k * 2] + n1 * r2[..., k * 2 + 1] for k in range(4)], axis=-1)
                min1, max1 = minmax_element(p1)
                min2, max2 = minmax_element(p2)
                sep = (max1 < min2) | (max2 < min1)
            collide &= ~sep
    return collide.astype(np.uint8)


def _quiet(fn):
    """numpy scalar arithmetic warns on overflow / invalid; non-finite inputs are part of the contract here."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        with np.errstate(all="ignore"):
            return fn(*a, **kw)
    return wrapped


@_quiet
def convex_collide_scalar(r1, r2):
    """Pure-Python loop form of utils.cu:159-184 for one pair (small cases)."""
    r1 = [F32(v) for v in r1]
    r2 = [F32(v) for v in r2]
    collide = 1
    for r in (r1, r2):
        for i in range(4):
            n0 = F32(r[(i + 1) * 2 % 8] - r[i * 2])
            n1 = F32(r[((i + 1) * 2 + 1) % 8] - r[i * 2 + 1])
            p1 = [F32(F32(n0 * r1[k * 2]) + F32(n1 * r1[k * 2 + 1])) for k in range(4)]
            p2 = [F32(F32(n0 * r2[k * 2]) + F32(n1 * r2[k * 2 + 1])) for k in range(4)]
            (min1, max1), (min2, max2) = _minmax_scalar(p1), _minmax_scalar(p2)
            if max1 < min2 or max2 < min1:
                collide = 0
    return collide


def _minmax_scalar(p):
    """thrust::minmax_element for a list of scalars (see minmax_element)."""
    lo = hi = p[0]
    for e in p[1:]:
        if e < lo:
            lo = e
        if hi < e:
            hi = e
    return lo, hi


@_quiet
def poly_collide(ax, ay, ka, bx, by, kb):
    """Convex polygon SAT with true normals for one pair;
    ax/ay/bx/by are float32 sequences, ka/kb the vertex counts."""
    A = [(F32(ax[i]), F32(ay[i])) for i in range(ka)]
    B = [(F32(bx[i]), F32(by[i])) for i in range(kb)]
    collide = 1
    for P in (A, B):
        kp = len(P)
        for i in range(kp):
            ex = F32(P[(i + 1) % kp][0] - P[i][0])
            ey = F32(P[(i + 1) % kp][1] - P[i][1])
            nx, ny = F32(-ey), ex
            p1 = [F32(F32(nx * x) + F32(ny * y)) for x, y in A]
            p2 = [F32(F32(nx * x) + F32(ny * y)) for x, y in B]
            (min1, max1), (min2, max2) = _minmax_scalar(p1), _minmax_scalar(p2)
            if max1 < min2 or max2 < min1:
                collide = 0
    return collide


def poly_collide_batch(vx, vy, k):
    """vx, vy: float32 [2][KMAX][n]; k: uint8 [2][n] -> uint8 [n] (vectorised
    by masking padded vertices with +/-inf in the min/max)."""
    vx = np.asarray(vx, dtype=F32)
    vy = np.asarray(vy, dtype=F32)
    k = np.asarray(k)
    n = vx.shape[-1]
    collide = np.ones(n, dtype=bool)
    vidx = np.arange(KMAX)[:, None]
    valid = [vidx < k[0][None, :], vidx < k[1][None, :]]
    for p in range(2):
        kp = k[p].astype(np.int64)
        for i in range(KMAX):
            has_edge = i < kp
            if not has_edge.any():
                break
            i1 = np.where(i + 1 < kp, i + 1, 0)
            cols = np.arange(n)
            with np.errstate(invalid="ignore", over="ignore"):
                ex = vx[p][i1, cols] - vx[p][i]
                ey = vy[p][i1, cols] - vy[p][i]
                nx, ny = -ey, ex
                proj = [nx[None, :] * vx[q] + ny[None, :] * vy[q] for q in range(2)]
                mn, mx = [], []
                for q in range(2):  # comparison-based extremes over the valid vertices, element 0 first (see minmax_element)
                    lo, hi = proj[q][0].copy(), proj[q][0].copy()
                    for r in range(1, KMAX):
                        e = proj[q][r]
                        lo = np.where(valid[q][r] & (e < lo), e, lo)
                        hi = np.where(valid[q][r] & (hi < e), e, hi)
                    mn.append(lo)
                    mx.append(hi)
                sep = (mx[0] < mn[1]) | (mx[1] < mn[0])
            collide &= ~(sep & has_edge)
    return collide.astype(np.uint8)


def place_polygon_given_cs(xs, ys, dx, dy, c, s):
    """rot_trans_rectangle (utils.cu:136-141) applied to every vertex of a polygon, cos / sin supplied by the caller:
    the robot placement of the polygon Monte-Carlo (ccp.cu:132-133)."""
    xs, ys = np.asarray(xs, dtype=F32), np.asarray(ys, dtype=F32)
    dx, dy, c, s = F32(dx), F32(dy), F32(c), F32(s)
    return (c * xs - s * ys) + dx, (s * xs + c * ys) + dy


def sample_polygon_given_cs(xs, ys, std_dev, normals5, c, s):
    """sample_rectangle (utils.cu:144-157) for a polygon about the origin: the five normals in the reference's order
    dx, dy, dtheta, dw, dh; dw, dh change the shape first — a rectangle's half extents grow by dw / 2, dh / 2
    (utils.cu:152-155), a polygon's obstacle-frame coordinates are scaled by (1 + dw), (1 + dh), std_dev width / height
    being relative — then the shape is rotated by dtheta (cos / sin supplied: the canonical sincos lives in the C oracle)
    and moved by (dx, dy) (utils.cu:156).  Written against the reference source and include/c2d.h, not against
    oracle/c2d_oracle.c."""
    xs, ys = np.asarray(xs, dtype=F32), np.asarray(ys, dtype=F32)
    sd = [F32(v) for v in std_dev]
    n = [F32(v) for v in normals5]
    with np.errstate(all="ignore"):
        dx, dy = n[0] * sd[0], n[1] * sd[1]
        dw, dh = n[3] * sd[3], n[4] * sd[4]
        fx, fy = F32(1) + dw, F32(1) + dh
        x, y = fx * xs, fy * ys
        c, s = F32(c), F32(s)
        return (c * x - s * y) + dx, (s * x + c * y) + dy


def _main():
    """BASELINE config 1: the 1 000 fixed OBB pairs of tests/golden/sat_rect_1k.npz on the CPU,
    boolean collide output.  `python oracle/sat.py [--print]`"""
    import os
    import sys
    import time

    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "..", "tests", "golden", "sat_rect_1k.npz"))
    planes = g["planes"]
    t0 = time.perf_counter()
    out = convex_collide(planes[:8].T, planes[8:].T)
    dt = time.perf_counter() - t0
    ok = np.array_equal(out, g["expected"])
    if "--print" in sys.argv:
        print("".join(str(int(v)) for v in out))
    print(f"{len(out)} pairs, {int(out.sum())} colliding, {dt * 1e3:.2f} ms (numpy float32), matches the golden vector: {ok}")
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(_main())
