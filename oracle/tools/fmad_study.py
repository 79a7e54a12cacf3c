#!/usr/bin/env python3
"""How far can the canonical (unfused) arithmetic be from a CUDA build of the reference?

TEST INFRASTRUCTURE (CPU only).  nvcc compiles the reference with -fmad=true by default, so each two-product sum
a*x + b*y of utils.cu:139-140 (rotation) and :173-174 (projection) is probably evaluated as one multiply and one
fused multiply-add; which product gets fused is ptxas' choice and cannot be observed without CUDA.  The oracle is
therefore built three times (oracle/Makefile): canonical, fma(a,x,b*y) ("fmad1") and fma(b,y,a*x) ("fmad2"), and
this script counts how many results differ between the canonical build and each fused build on

  A  BASELINE config 2 — 10^7 random OBB pairs (seed 0x5A7), vertices given (the SAT dot products only);
  B  the same pairs in pose format (rotation AND projection contracted);
  C  the razor-edge set of tests/test_gpu_sat.py (touching rectangles shifted by 0..+-2 ulp);
  D  the 1 000 golden pairs of BASELINE config 1 (tests/golden/sat_rect_1k.npz);
  E  BASELINE config 3 — Monte-Carlo hit counts of the bench scene at S samples (default 10^8).

usage: fmad_study.py [--pairs N] [--samples S] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import cpu as canon  # noqa: E402

load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")


def razor_edge_poses(n=20000):
    rng = np.random.default_rng(12)
    w1, h1, w2, h2 = (rng.uniform(0.5, 3, n).astype(np.float32) for _ in range(4))
    th = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    cx = rng.uniform(-50, 50, n).astype(np.float32)
    cy = rng.uniform(-50, 50, n).astype(np.float32)
    kk = (rng.integers(-8, 9, n) * 0.25).astype(np.float32)
    gap = ((w1 + w2) / 2 + kk * np.float32(2.0**-17)).astype(np.float32)
    cx2 = (cx + gap * np.cos(th)).astype(np.float32)
    cy2 = (cy + gap * np.sin(th)).astype(np.float32)
    return np.stack([cx, cy, w1, h1, th, cx2, cy2, w2, h2, th]).astype(np.float32)


def planes_of(mod, poses):
    return np.concatenate([mod.rects_from_poses(*poses[:5]), mod.rects_from_poses(*poses[5:])])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10_000_000)
    ap.add_argument("--samples", type=int, default=100_000_000)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    variants = {"fmad1: fma(a,x,b*y)": canon.load_variant("fmad1"), "fmad2: fma(b,y,a*x)": canon.load_variant("fmad2")}
    assert canon.lib().c2d_oracle_fmad_variant() == 0
    res = {"pairs": args.pairs, "samples": args.samples, "rows": []}

    def row(case, n, name, differ, extra=""):
        res["rows"].append({"case": case, "units": n, "variant": name, "differ": int(differ), "note": extra})
        print(f"{case:58s} {name:22s} {int(differ):>9d} of {n:<11d} {extra}")

    poses = wl.random_obb_pose_planes(args.pairs, seed=0x5A7)
    planes = planes_of(canon, poses)
    ref, ref_cnt = canon.sat_rect_pairs_verts(planes)
    ref_pose, _ = canon.sat_rect_pairs_pose(poses)
    assert np.array_equal(ref, ref_pose)
    print(f"config 2: {args.pairs} pairs, {ref_cnt} colliding under the canonical arithmetic")
    for name, m in variants.items():
        out, _ = m.sat_rect_pairs_verts(planes)
        row("A config 2, vertex format (projection contracted)", args.pairs, name, (out != ref).sum())
        vplanes = planes_of(m, poses)
        vbits = (vplanes.view(np.uint32) != planes.view(np.uint32)).sum()
        outp, _ = m.sat_rect_pairs_pose(poses)
        dmax = float(np.abs(vplanes.astype(np.float64) - planes.astype(np.float64)).max())
        row("B config 2, pose format (rotation + projection contracted)", args.pairs, name, (outp != ref).sum(),
            f"[{vbits} of {planes.size} vertex coordinates differ, largest |difference| {dmax:.2e}]")

    rp = razor_edge_poses()
    rplanes = planes_of(canon, rp)
    rref, _ = canon.sat_rect_pairs_verts(rplanes)
    for name, m in variants.items():
        out, _ = m.sat_rect_pairs_verts(rplanes)
        row("C razor-edge set, canonical vertices", rp.shape[1], name, (out != rref).sum())
        outp, _ = m.sat_rect_pairs_pose(rp)
        row("C razor-edge set, pose format", rp.shape[1], name, (outp != rref).sum())

    g = np.load(os.path.join(ROOT, "tests", "golden", "sat_rect_1k.npz"))
    gplanes = np.ascontiguousarray(g["planes"]) if "planes" in g else None
    if gplanes is not None:
        gref, _ = canon.sat_rect_pairs_verts(gplanes)
        for name, m in variants.items():
            out, _ = m.sat_rect_pairs_verts(gplanes)
            row("D config 1 golden pairs", gplanes.shape[1], name, (out != gref).sum())

    sc = wl.MC_PAIR_SCENE
    t0 = time.time()
    S = args.samples
    hits = {}
    for name, m in [("canonical", canon)] + list(variants.items()):
        hits[name] = m.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, S)
    print(f"config 3: {S} samples per build ({time.time() - t0:.0f} s); canonical hits {hits['canonical']} (p = {hits['canonical'] / S:.9f})")
    for name in variants:
        d = hits[name] - hits["canonical"]
        row("E config 3 hit count (same Philox stream)", S, name, abs(d), f"[net {d:+d} hits, p shifts by {d / S:+.2e}]")
    res["mc_hits"] = hits
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
