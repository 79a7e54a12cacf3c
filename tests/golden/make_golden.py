#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle.

The reference holds no golden vectors and cannot run here
(F1, F7), so these vectors are produced by this repo's own oracle
(oracle/c2d_oracle.c, cross-checked against oracle/sat.py) — "parity unpinned".
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu, sat  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def kat_pairs():
    """Known-answer rectangle pairs as (r1[8], r2[8], expected, name)."""
    def rect(cx, cy, w, h):
        return np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy - h / 2, cx + w / 2, cy + h / 2, cx - w / 2, cy + h / 2], np.float32)

    def diamond(cx, cy, r):
        return np.array([cx, cy - r, cx + r, cy, cx, cy + r, cx - r, cy], np.float32)

    K = []
    K.append((rect(0, 0, 2, 1), rect(0, 0, 2, 1), 1, "identical"))
    K.append((rect(0, 0, 2, 1), rect(10, 10, 2, 1), 0, "far apart"))
    K.append((rect(0, 0, 2, 2), rect(2, 0, 2, 2), 1, "edge touching (strict <)"))
    K.append((rect(0, 0, 2, 2), rect(2, 2, 2, 2), 1, "corner touching"))
    K.append((rect(0, 0, 2, 2), rect(2.5, 0, 2, 2), 0, "gap 0.5"))
    K.append((rect(0, 0, 2, 2), diamond(1.75, 1.75, 1.0), 0, "diamond: AABBs overlap, shapes do not"))
    K.append((rect(0, 0, 2, 2), diamond(1.4, 1.4, 1.0), 1, "diamond overlapping the corner"))
    K.append((rect(0, 0, 4, 4), rect(0.5, 0.25, 1, 1), 1, "contained"))
    K.append((rect(0, 0, 0, 2), rect(0.5, 0, 2, 2), 1, "zero-width rect (segment) inside"))
    K.append((rect(0, 0, 0, 2), rect(3, 0, 2, 2), 0, "zero-width rect outside"))
    K.append((rect(0, 0, -2, 1), rect(0.5, 0, 1, 1), 1, "negative width (after dw perturbation) overlapping"))
    K.append((rect(0, 0, -2, 1), rect(5, 0, 1, 1), 0, "negative width, apart"))
    K.append((rect(0, 0, 2, 2), rect(0, 2.0000002, 2, 2), 0, "one ulp gap in y"))
    K.append((rect(1e3, 1e3, 2, 2), rect(1e3 + 1, 1e3 + 1, 2, 2), 1, "large offset overlapping"))
    return K


def main():
    # ---- BASELINE config 1: 1 000 fixed OBB pairs ------------------------------------
    n = 1000
    kats = kat_pairs()
    poses = wl.random_obb_pose_planes(n, seed=1001, extent=4.0)  # denser than config 2: ~35 % collide
    r1 = cpu.rects_from_poses(*poses[:5])
    r2 = cpu.rects_from_poses(*poses[5:])
    planes = np.concatenate([r1, r2])  # [16][n]
    for i, (a, b, _, _) in enumerate(kats):
        planes[:8, i] = a
        planes[8:, i] = b
        poses[:, i] = np.nan  # KAT slots have no pose form
    expected, count = cpu.sat_rect_pairs_verts(planes)
    exp_np = sat.convex_collide(planes[:8].T, planes[8:].T)
    assert np.array_equal(expected, exp_np), "numpy and C oracles disagree"
    for i, (_, _, want, name) in enumerate(kats):
        assert expected[i] == want, f"KAT {name}: oracle says {expected[i]}, analytic answer {want}"
    np.savez_compressed(os.path.join(HERE, "sat_rect_1k.npz"), planes=planes, poses=poses, expected=expected,
                        n_kat=np.int64(len(kats)), kat_names=np.array([k[3] for k in kats]))
    print(f"sat_rect_1k.npz: {n} pairs, {count} colliding, {len(kats)} KATs")

    # ---- polygons K<=16 ----------------------------------------------------------------
    vx, vy, k = wl.random_convex_polygons(1000, seed=2002, extent=3.0)
    exp, cnt = cpu.sat_poly_pairs(vx, vy, k)
    assert np.array_equal(exp, sat.poly_collide_batch(vx, vy, k))
    np.savez_compressed(os.path.join(HERE, "poly_k16_1k.npz"), vx=vx, vy=vy, k=k, expected=exp)
    print(f"poly_k16_1k.npz: 1000 pairs, {cnt} colliding")

    # ---- random stream: raw Philox words and normals ---------------------------------------
    seed, scene = 0x0123456789ABCDEF, 0xFEDCBA9876543210
    begin = (1 << 31) - 6  # the block counter 8 * (sample >> 2) + b crosses its 32-bit carry inside the run
    raw = cpu.raw8(seed, scene, (1 << 33) - 4, 16)  # scene-sampler layout (blocks 2s, 2s + 1), crosses the carry too
    words = cpu.draw_words(seed, scene, begin, 16)
    nrm = cpu.normals5(seed, scene, begin, 16)
    np.savez_compressed(os.path.join(HERE, "philox_stream.npz"), seed=np.uint64(seed), scene=np.uint64(scene),
                        sample_begin=np.uint64(begin), raw=raw, raw_begin=np.uint64((1 << 33) - 4), draw_words=words, normals=nrm)
    print("philox_stream.npz: 16 samples")

    # ---- MC: fixed scenes, exact hit counts ---------------------------------------------------
    sc = wl.MC_PAIR_SCENE
    cases = []
    rng = np.random.default_rng(5)
    scenes = [(sc["pos"], sc["pose"], sc["std_dev"])]
    for _ in range(7):
        scenes.append(((float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))),
                       (float(rng.uniform(0.1, 5)), float(rng.uniform(0.1, 5)), float(rng.uniform(0, 6.28))),
                       tuple(float(v) for v in np.sqrt(rng.uniform(0, 0.3, 5)))))
    for sid, (pos, pose, sd) in enumerate(scenes):
        hits = cpu.mc_pair(sc["robot_w"], sc["robot_h"], pos, pose, sd, 1234, sid, 1000, 200000)
        cases.append((pos + pose + sd, sid, hits))
    np.savez_compressed(os.path.join(HERE, "mc_pair_cases.npz"),
                        params=np.array([c[0] for c in cases], np.float32),
                        scene_id=np.array([c[1] for c in cases], np.uint64),
                        hits=np.array([c[2] for c in cases], np.uint64),
                        seed=np.uint64(1234), sample_begin=np.uint64(1000), n_samples=np.uint64(200000),
                        robot=np.array([sc["robot_w"], sc["robot_h"]], np.float32))
    print("mc_pair_cases.npz:", [c[2] for c in cases])

    # ---- MC adaptive: 64 scenes, max_samples 25 000 ------------------------------------------------
    poses_t, sd_t, _ = wl.random_tables(32, 32, seed=11, shape_variance=True)
    scn = cpu.sample_scenes(poses_t, sd_t, sc["robot_w"], sc["robot_h"], 4.0, 99, 0, 64)
    hits, used, rows, total = cpu.mc_scenes(poses_t, sd_t, scn, sc["robot_w"], sc["robot_h"], wl.DEFAULT_BINS,
                                            wl.DEFAULT_BIN_ACCURACY, 25000, 4321, 0)
    np.savez_compressed(os.path.join(HERE, "mc_scenes_64.npz"), poses=poses_t, std_devs=sd_t, scenes=scn, hits=hits,
                        n_used=used, rows=rows, total=np.uint64(total), max_samples=np.uint32(25000), seed=np.uint64(4321),
                        scene_seed=np.uint64(99), spread=np.float32(4.0))
    print("mc_scenes_64.npz: total samples", total, "n_used histogram", np.unique(used, return_counts=True))

    # ---- MC over convex polygons: fixed scenes, exact hit counts (oracle/c2d_oracle.c "Monte-Carlo over convex polygons") -------------
    rng = np.random.Generator(np.random.Philox(2024))
    rec = {k: [] for k in ("ka", "kb", "rx", "ry", "ox", "oy", "pos", "theta", "std_dev", "seed", "scene", "begin", "n", "hits")}
    specs = [(7, 5, wl.mc_poly_pair_scene()["pos"], (0.3, 0.3, 0.2, 0.0, 0.0)), (3, 3, (1.8, 0.4), (0.4, 0.2, 0.3, 0.1, 0.0)), (16, 16, (3.2, -1.0), (0.3, 0.3, 0.2, 0.05, 0.08)),
             (4, 12, (-2.6, 1.4), (0.5, 0.5, 0.4, 0.0, 0.0)), (9, 2, (2.2, 0.3), (0.2, 0.6, 1.0, 0.0, 0.3)), (1, 6, (0.4, 0.2), (0.8, 0.8, 0.0, 0.2, 0.2)),
             (5, 8, (6.5, 2.0), (0.4, 0.4, 0.2, 0.0, 0.0)), (12, 7, (9.0, -9.0), (0.5, 0.5, 0.5, 0.1, 0.1))]
    for i, (ka, kb, pos, sd) in enumerate(specs):
        scp = wl.mc_poly_pair_scene(ka, kb, seed=300 + i) if i else wl.mc_poly_pair_scene()
        theta = 0.6 if i == 0 else float(rng.uniform(-3.2, 3.2))  # case 0: the bench scene of the polygon leg
        begin, n = 1000 + 3 * i, 150_000
        h = cpu.mc_poly_pair(scp["robot"], pos, theta, scp["obstacle"], sd, 1234, i, begin, n)
        pad = lambda v: np.concatenate([v, np.zeros(wl.KMAX - len(v), np.float32)])  # noqa: E731
        for key, val in (("ka", ka), ("kb", kb), ("rx", pad(scp["robot"][0])), ("ry", pad(scp["robot"][1])), ("ox", pad(scp["obstacle"][0])),
                         ("oy", pad(scp["obstacle"][1])), ("pos", pos), ("theta", theta), ("std_dev", sd), ("seed", 1234), ("scene", i), ("begin", begin),
                         ("n", n), ("hits", h)):
            rec[key].append(val)
    np.savez_compressed(os.path.join(HERE, "mc_poly_pair_cases.npz"), ka=np.array(rec["ka"], np.uint32), kb=np.array(rec["kb"], np.uint32),
                        rx=np.array(rec["rx"], np.float32), ry=np.array(rec["ry"], np.float32), ox=np.array(rec["ox"], np.float32), oy=np.array(rec["oy"], np.float32),
                        pos=np.array(rec["pos"], np.float32), theta=np.array(rec["theta"], np.float32), std_dev=np.array(rec["std_dev"], np.float32),
                        seed=np.array(rec["seed"], np.uint64), scene=np.array(rec["scene"], np.uint64), begin=np.array(rec["begin"], np.uint64),
                        n=np.array(rec["n"], np.uint64), hits=np.array(rec["hits"], np.uint64))
    print("mc_poly_pair_cases.npz:", rec["hits"])

    # ---- MC over convex polygons, adaptive: 48 scenes, max_samples 25 000 -------------------------------------------------------------
    pposes, psd = wl.random_poly_tables(24, 24, seed=13, shape_variance=True)
    scp = wl.mc_poly_pair_scene(9, 5, seed=14)
    pscn = wl.random_poly_scenes(48, pposes, psd, 2.3, seed=15)
    hits, used, rows, total = cpu.mc_poly_scenes(scp["robot"], pposes, psd, pscn, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 25000, 4321, 500)
    np.savez_compressed(os.path.join(HERE, "mc_poly_scenes_48.npz"), robot_x=scp["robot"][0], robot_y=scp["robot"][1], poly_poses=pposes, std_devs=psd,
                        scenes=pscn, hits=hits, n_used=used, rows=rows, total_samples=np.uint64(total), max_samples=np.uint32(25000), seed=np.uint64(4321),
                        scene_id_base=np.uint64(500))
    print("mc_poly_scenes_48.npz: total samples", total, "n_used histogram", np.unique(used, return_counts=True))


if __name__ == "__main__":
    main()
