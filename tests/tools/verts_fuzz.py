#!/usr/bin/env python3
"""Differential fuzz of the VERTEX-format pair entry points — the headline path, c2d_sat_rect_pairs_verts (utils.cu:159-184 per
pair), and its two siblings c2d_sat_rect_pairs_verts_mask and c2d_sat_rect_pairs_aos — against the CPU oracle.  Shapes: rectangles
built from poses at one scale or at a random power of ten (1e-14 .. 1e14) and at a distance from the origin, pairs built to
touch, general convex quadrilaterals (the kernels take ANY 16 floats per pair and certify four of the eight axes), and
non-finite coordinates.  Calls: batch sizes on and around the kernels' lane / wave / block edges, planes shifted 0..3 floats off
their 16-byte alignment together or one plane alone (4-pairs-per-lane path, one-pair path, and the seam between them), the output
0..3 bytes off its alignment, with a count that starts anywhere, without one.  Every boolean, every count and every byte behind
the output must equal the oracle's.  TEST INFRASTRUCTURE (uses oracle/).   usage: verts_fuzz.py [configs] [seed]"""
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

SIZES = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4097]


def convex_quads(n, rng, scale, offset):
    """[8][n]: four points on a random ellipse in angular order (half of them clockwise), rotated, scaled and moved"""
    ang = np.sort(rng.uniform(0, 2 * np.pi, (4, n)), axis=0)
    a, b = rng.uniform(0.3, 2.5, n), rng.uniform(0.3, 2.5, n)
    rot = rng.uniform(0, 2 * np.pi, n)
    cx, cy = rng.uniform(-3, 3, n) + offset, rng.uniform(-3, 3, n) + offset
    x, y = a * np.cos(ang), b * np.sin(ang)
    X = (np.cos(rot) * x - np.sin(rot) * y + cx) * scale
    Y = (np.sin(rot) * x + np.cos(rot) * y + cy) * scale
    cw = rng.random(n) < 0.5
    X[:, cw], Y[:, cw] = X[::-1][:, cw], Y[::-1][:, cw]
    q = np.empty((8, n), np.float32)
    q[0::2], q[1::2] = X, Y
    return q


def make(rng, n):
    """planes float32 [16][n] and what they are"""
    kind = ["rects", "rects", "touching", "quads", "non-finite"][int(rng.integers(5))]
    scale = float(10.0 ** rng.integers(-14, 15)) if rng.random() < 0.4 else 1.0
    offset = float(rng.choice([0.0, 0.0, 3.0, 100.0, 1e4]))
    with np.errstate(all="ignore"):
        if kind == "quads":
            planes = np.concatenate([convex_quads(n, rng, scale, offset), convex_quads(n, rng, scale, offset)])
        else:
            if kind == "touching":
                poses = wl.touching_pose_pairs(n, seed=int(rng.integers(1 << 30)), scale=scale, offset=offset)
            else:
                poses = wl.random_obb_pose_planes(n, seed=int(rng.integers(1 << 30)), extent=float(rng.choice([0.5, 2.0, 8.0])))
                poses[[0, 1, 5, 6]] += np.float32(offset)
                poses[[0, 1, 2, 3, 5, 6, 7, 8]] *= np.float32(scale)
            planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
            if kind == "non-finite":
                planes = wl.inject_non_finite(planes, seed=int(rng.integers(1 << 30)), frac=0.2)
    return np.ascontiguousarray(planes, np.float32), (kind, scale, offset)


def one(eng, rng, idx, announce=None):
    """One configuration; `announce(text)` is called with its description BEFORE any GPU work, so that a fault names its input."""
    n = int(rng.choice(SIZES + [int(rng.integers(1, 5000)), int(rng.integers(5000, 400_000))]))
    planes, what = make(rng, n)
    entry = ["verts", "verts", "mask", "aos"][int(rng.integers(4))]
    shift = int(rng.integers(0, 4)) if rng.random() < 0.5 else 0           # all planes off their 16-byte alignment by `shift` floats
    lone = int(rng.integers(0, 16)) if rng.random() < 0.2 else -1           # ... or one plane alone (by one float)
    out_off = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
    counted = rng.random() < 0.7
    count0 = int(rng.integers(0, 1 << 40)) if counted and rng.random() < 0.5 else 0
    if announce is not None:
        announce(f"config {idx}: {entry} n {n} {what} plane shift {shift} floats, lone plane {lone}, output offset {out_off} bytes, "
                 f"{'count from %d' % count0 if counted else 'no count'}")
    with np.errstate(all="ignore"):
        ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    d_cnt = eng.to_device(np.array([count0], np.uint64)) if counted else None
    bufs = [d_cnt] if counted else []
    if entry == "aos":
        d1, d2 = eng.to_device(np.ascontiguousarray(planes[:8].T)), eng.to_device(np.ascontiguousarray(planes[8:].T))
        d_out = eng.to_device(np.full(n + 16, 0xA5, np.uint8))
        eng.sat_rect_pairs_aos(d1, d2, n, d_out.ptr + out_off, d_cnt)
        raw = d_out.get()
        got, clean = raw[out_off:out_off + n], (raw[:out_off] == 0xA5).all() and (raw[out_off + n:] == 0xA5).all()
        bufs += [d1, d2, d_out]
    else:
        host = np.zeros((16, n + 8), np.float32)
        offs = [shift + (1 if k == lone else 0) for k in range(16)]
        for k in range(16):
            host[k, offs[k]:offs[k] + n] = planes[k]
        d = eng.to_device(host)
        ptrs = [d.row(k) + 4 * offs[k] for k in range(16)]
        bufs.append(d)
        if entry == "mask":
            words = (n + 63) // 64
            d_mask = eng.to_device(np.full(words + 2, 0xA5A5A5A5A5A5A5A5, np.uint64))
            eng.sat_rect_pairs_verts_mask(ptrs, n, d_mask, d_cnt)
            mask = d_mask.get()
            bits = np.unpackbits(mask[:words].view(np.uint8), bitorder="little")
            got, clean = bits[:n], (mask[words:] == 0xA5A5A5A5A5A5A5A5).all() and not bits[n:].any()
            bufs.append(d_mask)
        else:
            d_out = eng.to_device(np.full(n + 16, 0xA5, np.uint8))
            eng.sat_rect_pairs_verts(ptrs, n, d_out.ptr + out_off, d_cnt)
            raw = d_out.get()
            got, clean = raw[out_off:out_off + n], (raw[:out_off] == 0xA5).all() and (raw[out_off + n:] == 0xA5).all()
            bufs.append(d_out)
    cnt_ok = (not counted) or int(d_cnt.get()[0]) == count0 + ref_cnt
    bad = int((got != ref).sum())
    ok = bad == 0 and bool(clean) and cnt_ok
    for a in bufs:
        a.free()
    if not ok:
        print(f"MISMATCH config {idx}: {entry} n {n} {what} shift {shift} lone {lone} out_off {out_off}: {bad} booleans differ, bytes around the output "
              f"{'untouched' if clean else 'WRITTEN'}, count {'ok' if cnt_ok else 'WRONG'}", flush=True)
    return ok, (entry, what[0], n)


def main():
    configs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
    eng = pkg.Engine(0)
    verbose = bool(os.environ.get("C2D_FUZZ_VERBOSE"))
    fails, pairs = 0, 0
    for i in range(configs):
        ok, info = one(eng, rng, i, (lambda t: print(t, flush=True)) if verbose else None)
        fails += not ok
        pairs += info[2]
        if (i + 1) % 100 == 0:
            print(f"  {i + 1} / {configs} configurations, {pairs} pairs, {fails} failures so far", flush=True)
    print(f"verts fuzz: {configs} configurations, {pairs} pairs, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
