"""CPU tests of the oracle's Monte-Carlo over convex polygons (oracle/c2d_oracle.c, "Monte-Carlo over convex polygons"): the
restatement the GPU kernels of csrc/c2d_mc_poly.hip are compared with.  The reference has no polygon sampler (README.md:3 only
says the code "can easily be extended"), so the extension is pinned here by (i) an independent numpy restatement written against
utils.cu:144-157 (oracle/sat.py), (ii) the rectangle case, where it must reproduce the rectangle oracle, (iii) closed forms,
(iv) committed golden vectors."""
import math
import os

import numpy as np
from scipy.stats import norm

from oracle import sat

GOLD = os.path.join(os.path.dirname(__file__), "golden")
W, H = 4.07, 1.74


def test_numpy_and_c_agree_on_sampled_polygons_and_their_test(oracle, wl):
    """sample by sample: the sampled obstacle's vertices and the collide result, numpy against C, bit for bit"""
    rng = np.random.Generator(np.random.Philox(3))
    checked = 0
    for trial in range(24):
        ka, kb = int(rng.integers(1, 17)), int(rng.integers(1, 17))
        robot = wl.convex_polygon(ka, rng, rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0, 6.28))
        obstacle = wl.convex_polygon(kb, rng, rng.uniform(0.2, 2), rng.uniform(0.2, 2), rng.uniform(0, 6.28), clockwise=bool(trial & 1))
        pos, theta = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))), float(rng.uniform(-3.2, 3.2))
        sd = tuple(float(v) for v in rng.uniform(0, 0.5, 5) * np.array([1, 1, 1, trial % 2, trial % 3 == 0]))
        s_, c_ = oracle.sincosf(theta)
        rx, ry = sat.place_polygon_given_cs(robot[0], robot[1], pos[0], pos[1], c_[0], s_[0])
        crx, cry = oracle.place_polygon(robot, pos[0], pos[1], theta)
        assert np.array_equal(rx.view(np.uint32), crx.view(np.uint32)) and np.array_equal(ry.view(np.uint32), cry.view(np.uint32))
        n = 40
        normals = oracle.normals5(13, trial, 1000, n)
        hits = 0
        for i in range(n):
            sn, cs = oracle.sincosf(np.float32(normals[i][2]) * np.float32(sd[2]))
            ox, oy = sat.sample_polygon_given_cs(obstacle[0], obstacle[1], sd, normals[i], cs[0], sn[0])
            cox, coy = oracle.mc_poly_sampled(obstacle, sd, 13, trial, 1000 + i)
            assert np.array_equal(ox.view(np.uint32), cox.view(np.uint32)) and np.array_equal(oy.view(np.uint32), coy.view(np.uint32))
            hits += sat.poly_collide(rx, ry, ka, ox, oy, kb)
            checked += 1
        assert hits == oracle.mc_poly_pair(robot, pos, theta, obstacle, sd, 13, trial, 1000, n)
    assert checked == 24 * 40


def test_rectangles_as_polygons_reproduce_the_rectangle_oracle(oracle, wl):
    """sigma_w = sigma_h = 0: a rectangle as a 4-gon gets the rectangle path's vertices, sample for sample; axis-aligned scenes
    give identical hit counts (the two tests then scale the same coordinate differences by different positive constants only),
    rotated scenes may differ for a sample within an ulp of touching"""
    robot, obstacle = wl.rect_polygon(W, H), wl.rect_polygon(2.0, 1.0)
    sd = (0.3, 0.3, 0.2, 0.0, 0.0)
    for s in range(50):
        ox, oy = oracle.mc_poly_sampled(obstacle, sd, 1234, 0, s)
        r = oracle.mc_sampled_rect((2.0, 1.0, 0.6), sd, 1234, 0, s)
        assert np.array_equal(np.stack([ox, oy], 1).reshape(-1).view(np.uint32), r.view(np.uint32))
    rr = np.array(oracle.create_rect(W, H))
    rr = oracle.rot_trans_rectangle(rr, 3.0, 1.0, 0.6)
    px, py = oracle.place_polygon(robot, 3.0, 1.0, 0.6)
    assert np.array_equal(np.stack([px, py], 1).reshape(-1).view(np.uint32), rr.view(np.uint32))
    n = 2_000_000
    for pos, theta, sdv, seed in [((3.0, 1.0), 0.0, (0.3, 0.3, 0.0, 0, 0), 1), ((2.5, 0.2), 0.0, (0.5, 0.2, 0.0, 0, 0), 2)]:
        assert oracle.mc_poly_pair(robot, pos, theta, obstacle, sdv, seed, 0, 0, n) == oracle.mc_pair(W, H, pos, (2.0, 1.0, theta), sdv, seed, 0, 0, n)
    a = oracle.mc_poly_pair(robot, (3.0, 1.0), 0.6, obstacle, sd, 1234, 0, 0, n)
    b = oracle.mc_pair(W, H, (3.0, 1.0), (2.0, 1.0, 0.6), sd, 1234, 0, 0, n)
    assert abs(a - b) <= 3, (a, b)


def test_relative_shape_noise_has_the_rectangle_distribution(oracle, wl):
    """a w x h box as a polygon with sigma_w / w, sigma_h / h against the rectangle oracle with sigma_w, sigma_h: different
    arithmetic, same distribution (utils.cu:152-155) — the probabilities agree within their sampling error"""
    w, h = 2.0, 1.0
    sd_rect = (0.25, 0.2, 0.15, 0.3, 0.2)
    sd_poly = (0.25, 0.2, 0.15, 0.3 / w, 0.2 / h)
    n = 1_000_000
    pr = oracle.mc_pair(W, H, (3.1, 0.9), (w, h, 0.5), sd_rect, 5, 0, 0, n) / n
    pp = oracle.mc_poly_pair(wl.rect_polygon(W, H), (3.1, 0.9), 0.5, wl.rect_polygon(w, h), sd_poly, 6, 0, 0, n) / n
    assert abs(pr - pp) < 5 * math.sqrt(2 * pr * (1 - pr) / n), (pr, pp)


def test_closed_forms(oracle, wl):
    robot = wl.rect_polygon(W, H)
    w, h = 2.0, 1.0
    ox = np.array([-1, 0, 1, 1, 1, 0, -1, -1], np.float32) * np.float32(w / 2)  # a box with extra vertices on its edges
    oy = np.array([-1, -1, -1, 0, 1, 1, 1, 0], np.float32) * np.float32(h / 2)
    n = 400_000
    px, sx = 3.4, 0.5
    p = norm.cdf((px + (W + w) / 2) / sx) - norm.cdf((px - (W + w) / 2) / sx)
    got = oracle.mc_poly_pair(robot, (px, 0.0), 0.0, (ox, oy), (sx, 0, 0, 0, 0), 7, 0, 0, n) / n
    assert abs(got - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4
    px, srel = 3.3, 0.25
    a = (px - W / 2) / (w / 2)
    p = (1 - norm.cdf((a - 1) / srel)) + norm.cdf((-a - 1) / srel)
    got = oracle.mc_poly_pair(robot, (px, 0.0), 0.0, (ox, oy), (0, 0, 0, srel, 0), 8, 1, 0, n) / n
    assert abs(got - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4
    # a point obstacle (k = 1) with isotropic position noise against a disk-like 16-gon robot: p = P(|N| < R) within the polygon's
    # inscribed / circumscribed radii
    rng = np.random.Generator(np.random.Philox(1))
    ang = 2 * np.pi * np.arange(16) / 16
    disk = (np.cos(ang).astype(np.float32) * 2, np.sin(ang).astype(np.float32) * 2)
    got = oracle.mc_poly_pair(disk, (0.0, 0.0), 0.0, ([0.0], [0.0]), (1.0, 1.0, 0, 0, 0), 9, 2, 0, n) / n
    lo, hi = 1 - math.exp(-(2 * math.cos(math.pi / 16)) ** 2 / 2), 1 - math.exp(-2.0 ** 2 / 2)
    assert lo - 3e-3 < got < hi + 3e-3, (lo, got, hi)


def test_golden_vectors_and_range_additivity(oracle):
    g = np.load(os.path.join(GOLD, "mc_poly_pair_cases.npz"))
    for i in range(len(g["hits"])):
        ka, kb = int(g["ka"][i]), int(g["kb"][i])
        args = ((g["rx"][i][:ka], g["ry"][i][:ka]), tuple(g["pos"][i]), float(g["theta"][i]), (g["ox"][i][:kb], g["oy"][i][:kb]), tuple(g["std_dev"][i]),
                int(g["seed"][i]), int(g["scene"][i]))
        b, n = int(g["begin"][i]), int(g["n"][i])
        assert oracle.mc_poly_pair(*args, b, n) == int(g["hits"][i])
        assert oracle.mc_poly_pair(*args, b, 7001) + oracle.mc_poly_pair(*args, b + 7001, n - 7001) == int(g["hits"][i])


def test_adaptive_golden_subset(oracle, pkg):
    g = np.load(os.path.join(GOLD, "mc_poly_scenes_48.npz"))
    poses = g["poly_poses"].view(pkg.POLY_POSE_DT).reshape(-1)
    sds = g["std_devs"].view(pkg.STD_DT).reshape(-1)
    scenes = g["scenes"].view(pkg.SCENE_DT).reshape(-1)
    robot = (g["robot_x"], g["robot_y"])
    idx = np.argsort(g["n_used"])[:8]
    for i in idx:  # scene ids are positional: evaluate each selected scene with its own id
        hits, used, rows, _ = oracle.mc_poly_scenes(robot, poses, sds, scenes[i:i + 1], [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], int(g["max_samples"]),
                                                    int(g["seed"]), int(g["scene_id_base"]) + int(i))
        assert hits[0] == g["hits"][i] and used[0] == g["n_used"][i]
        assert rows.view(np.uint32).tolist() == g["rows"].view(np.uint32).reshape(len(scenes), -1)[i].tolist()


def test_bad_vertex_counts(oracle, wl):
    import pytest

    sc = wl.mc_poly_pair_scene()
    bad = oracle.polygon(*sc["robot"])
    bad.k = 17
    with pytest.raises(ValueError):
        oracle.mc_poly_pair(bad, sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1, 0, 0, 10)
