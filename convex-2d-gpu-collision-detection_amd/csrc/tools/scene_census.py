#!/usr/bin/env python3
"""Developer tool: where do the samples of the config-4 workload go?  Runs c2d_mc_scenes on n scenes, restates make_scene's
radius threshold (c2d_mc.hip) in numpy per scene — f = the fraction of radius words that are candidates, 1 when the
radius test does not apply — and prints how the drawn samples distribute over f, with each bucket's collision rate.
usage: scene_census.py [n_scenes] [max_samples]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
import importlib  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
KN = 6.77


def candidate_fraction(rw, rh, px, py, pose, sd):
    th = pose["theta"].astype(np.float64)
    c, s = np.cos(th), np.sin(th)
    hx, hy = rw / 2, rh / 2
    corners = np.stack([np.stack([c * sx * hx - s * sy * hy + px, s * sx * hx + c * sy * hy + py], -1)
                        for sx, sy in ((-1, -1), (1, -1), (1, 1), (-1, 1))], 1)  # [n][4][2]
    hxm = np.abs(pose["width"] / 2) + 0.5 * KN * np.abs(sd["width"])
    hym = np.abs(pose["height"] / 2) + 0.5 * KN * np.abs(sd["height"])
    rho = np.sqrt(hxm**2 + hym**2)
    D = KN * (np.abs(sd["x"]) + np.abs(sd["y"]))
    R0 = np.zeros(len(px))
    for i in range(2):
        a = corners[:, i + 1] - corners[:, i]
        proj = np.einsum("nk,nvk->nv", a, corners)
        n2, n1 = np.hypot(a[:, 0], a[:, 1]), np.abs(a).sum(1)
        M = n2 * rho * (1 + 2.0**-10) + 2.0**-12 * n1 * (rho + D)
        hi, lo = proj.max(1) + M, proj.min(1) - M
        G = np.hypot(a[:, 0] * sd["x"], a[:, 1] * sd["y"]) * (1 + 2.0**-10)
        L = np.where(lo > 0, lo, np.where(hi < 0, -hi, 0.0))
        R0 = np.maximum(R0, np.where((L > 0) & (G > 0), L / np.maximum(G, 1e-30), 0.0))
    f = np.where(R0 > 0.25, np.exp(-0.5 * (R0 * (1 - 2.0**-10)) ** 2), 1.0)
    return np.minimum(f, 1.0)


def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    max_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 120_000
    e0 = pkg.Engine(0)
    tp, ts, _ = wl.random_tables(65536, 65536, seed=7)
    d_p, d_s = e0.to_device(tp), e0.to_device(ts)
    d_sc = e0.empty(ns, pkg.SCENE_DT)
    e0.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, 7, 0, ns, d_sc)
    d_h, d_u = e0.zeros(ns, np.uint32), e0.zeros(ns, np.uint32)
    e0.mc_scenes(d_p, 65536, d_s, 65536, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, max_samples, 11, 0, d_h, d_u)
    e0.synchronize()
    sc, used, hits = d_sc.get(), d_u.get().astype(np.float64), d_h.get().astype(np.float64)
    pose = tp[sc["pose_idx"].astype(np.int64)]
    sd = ts[sc["var_idx"].astype(np.int64)]
    f = candidate_fraction(4.07, 1.74, sc["x"].astype(np.float64), sc["y"].astype(np.float64), pose, sd)
    edges = [0, 1e-6, 1e-4, 1e-3, 1e-2, 0.05, 0.25, 0.5, 0.999999, 1.0000001]
    tot = used.sum()
    print(f"{ns} scenes, {tot:.4g} samples; candidate fraction f of the radius test -> share of scenes / of samples / collision rate")
    for a, b in zip(edges[:-1], edges[1:]):
        m = (f >= a) & (f < b)
        if m.any():
            print(f"  f in [{a:g}, {b:g}): scenes {m.mean() * 100:5.1f} %   samples {used[m].sum() / tot * 100:5.1f} %   p = {hits[m].sum() / max(used[m].sum(), 1):.4f}"
                  f"   mean samples/scene {used[m].mean():.0f}")


if __name__ == "__main__":
    main()
