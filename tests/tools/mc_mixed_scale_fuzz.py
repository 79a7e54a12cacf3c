#!/usr/bin/env python3
"""Differential fuzz of c2d_mc_pair / c2d_mc_poly_pair against the CPU oracle on single scenes whose length parameters each get their OWN
random scale (1e-14 .. 1e7, some exactly zero): the shortcuts' margins are relative to sums of lengths, so a scene that mixes scales is where
an absolute rounding term would show.  TEST INFRASTRUCTURE (uses oracle/).   usage: mc_mixed_scale_fuzz.py [seed] [scenes]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle
eng = pkg.Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
def ex(lo=-14, hi=7):
    return float(10.0 ** rng.uniform(lo, hi)) * (1 if rng.random() < 0.9 else 0)
bad = 0
for i in range(N):
    # rectangles
    w, h = ex() or 1.0, ex() or 1.0
    pos = (ex() * rng.choice([-1, 1]), ex() * rng.choice([-1, 1]))
    pose = (ex() or 1.0, ex() or 1.0, float(rng.uniform(-7, 7)))
    sd = (ex(), ex(), float(rng.uniform(0, 1)) * (rng.random() < 0.8), ex(-14, 2) * (rng.random() < 0.3), ex(-14, 2) * (rng.random() < 0.3))
    if rng.random() < 0.5:  # make it interesting: put the robot within reach of the obstacle
        s = max(w, h, pose[0], pose[1], sd[0], sd[1])
        pos = (float(rng.uniform(-2, 2)) * s, float(rng.uniform(-2, 2)) * s)
    with np.errstate(all="ignore"):
        ref = oracle.mc_pair(w, h, pos, pose, sd, 3, i, 5, 20_001)
    d = eng.zeros(1, np.uint64); eng.mc_pair(w, h, pos, pose, sd, 3, i, 5, 20_001, d); got = int(d.get()[0]); d.free()
    if got != ref:
        bad += 1; print("RECT", i, "w,h", w, h, "pos", pos, "pose", pose, "sd", sd, "gpu", got, "oracle", ref, flush=True)
    # polygons
    ka, kb = int(rng.integers(1, 17)), int(rng.integers(1, 17))
    prng = np.random.Generator(np.random.Philox(int(rng.integers(1 << 30))))
    sr, so = ex() or 1.0, ex() or 1.0
    robot = wl.convex_polygon(ka, prng, sr * float(rng.uniform(0.3, 3)), sr * float(rng.uniform(0.3, 3)), float(rng.uniform(0, 6.28)), clockwise=bool(rng.integers(2)))
    obst = wl.convex_polygon(kb, prng, so * float(rng.uniform(0.3, 3)), so * float(rng.uniform(0.3, 3)), float(rng.uniform(0, 6.28)), clockwise=bool(rng.integers(2)))
    sdp = (ex(), ex(), float(rng.uniform(0, 1)) * (rng.random() < 0.8), float(rng.uniform(0, 0.2)) * (rng.random() < 0.3), float(rng.uniform(0, 0.2)) * (rng.random() < 0.3))
    s = max(sr, so, sdp[0], sdp[1])
    ppos = (float(rng.uniform(-3, 3)) * s, float(rng.uniform(-3, 3)) * s) if rng.random() < 0.7 else (ex() * rng.choice([-1, 1]), ex() * rng.choice([-1, 1]))
    th = float(rng.uniform(-7, 7))
    with np.errstate(all="ignore"):
        ref = oracle.mc_poly_pair(robot, ppos, th, obst, sdp, 3, i, 5, 20_001)
    d = eng.zeros(1, np.uint64); eng.mc_poly_pair(robot, ppos, th, obst, sdp, 3, i, 5, 20_001, d); got = int(d.get()[0]); d.free()
    if got != ref:
        bad += 1; print("POLY", i, "ka,kb", ka, kb, "scales", sr, so, "pos", ppos, "sd", sdp, "gpu", got, "oracle", ref, flush=True)
    if (i + 1) % 100 == 0: print(i + 1, "scenes of each kind,", bad, "differences so far", flush=True)
print(N, "scenes of each kind,", bad, "differences")
