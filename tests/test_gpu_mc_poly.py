"""GPU parity tests of the Monte-Carlo kernels for convex polygons (c2d_mc_poly_pair, c2d_mc_poly_scenes) through the C-ABI:
hit counts, stop points and output rows must equal the CPU oracle's restatement (oracle/c2d_oracle.c, "Monte-Carlo over convex
polygons": sample_rectangle utils.cu:144-157 and the loop of compute_collision_probability.cu:119-139 generalised, the
interval test of utils.cu:172-180 on true normals) bit for bit; probabilities are checked against closed forms with the
tolerance BASELINE.json states (1e-3 at 1e8 samples)."""
import math
import os

import numpy as np
import pytest
from scipy.stats import norm

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
W, H = 4.07, 1.74


def gpu_hits(eng, robot, pos, theta, obstacle, sd, seed, scene, begin, n):
    d = eng.zeros(1, np.uint64)
    eng.mc_poly_pair(robot, pos, theta, obstacle, sd, seed, scene, begin, n, d)
    h = int(d.get()[0])
    d.free()
    return h


def test_mc_poly_pair_golden_cases(eng):
    g = np.load(os.path.join(GOLD, "mc_poly_pair_cases.npz"))
    for i in range(len(g["hits"])):
        ka, kb = int(g["ka"][i]), int(g["kb"][i])
        robot, obstacle = (g["rx"][i][:ka], g["ry"][i][:ka]), (g["ox"][i][:kb], g["oy"][i][:kb])
        got = gpu_hits(eng, robot, tuple(g["pos"][i]), float(g["theta"][i]), obstacle, tuple(g["std_dev"][i]), int(g["seed"][i]), int(g["scene"][i]),
                       int(g["begin"][i]), int(g["n"][i]))
        assert got == int(g["hits"][i]), i


@pytest.mark.parametrize("ka,kb", [(1, 1), (2, 3), (3, 3), (4, 4), (5, 8), (7, 5), (8, 9), (12, 12), (13, 4), (16, 16), (16, 3), (3, 16)])
def test_mc_poly_pair_matches_oracle(eng, oracle, wl, ka, kb):
    """every vertex-count class of the evaluation (4 / 8 / 12 / 16 obstacle slots), unaligned sample ranges, with and without shape noise"""
    sc = wl.mc_poly_pair_scene(ka, kb, seed=100 + 17 * ka + kb)
    for sd, begin, n in [(sc["std_dev"], 0, 200_000), ((0.3, 0.25, 0.2, 0.1, 0.05), 5, 100_003), ((0.0, 0.0, 0.5, 0.0, 0.2), 2**40 + 3, 65_537)]:
        ref = oracle.mc_poly_pair(sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sd, 77, 3, begin, n)
        got = gpu_hits(eng, sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sd, 77, 3, begin, n)
        assert got == ref, (ka, kb, sd, begin, n, got, ref)


def test_mc_poly_pair_far_and_near_scenes(eng, oracle, wl):
    """scenes across the regimes of the sample loops: certain misses from the radius word (FAR), from the centre (NEAR with pretest),
    p ~ 0.5, certain hits; clockwise polygons"""
    rng = np.random.Generator(np.random.Philox(5))
    for i in range(40):
        ka, kb = int(rng.integers(3, 17)), int(rng.integers(3, 17))
        robot = wl.convex_polygon(ka, rng, rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(0, 6.28), clockwise=bool(i & 1))
        obstacle = wl.convex_polygon(kb, rng, rng.uniform(0.2, 2), rng.uniform(0.2, 2), rng.uniform(0, 6.28), clockwise=bool(i & 2))
        dist = [0.0, 1.5, 3.0, 4.5, 6.0, 9.0, 15.0][i % 7]
        ang = rng.uniform(0, 6.28)
        pos = (float(dist * np.cos(ang)), float(dist * np.sin(ang)))
        sd = (float(rng.uniform(0, 0.6)), float(rng.uniform(0, 0.6)), float(rng.uniform(0, 0.5)), float(rng.uniform(0, 0.1)) * (i % 3 == 0),
              float(rng.uniform(0, 0.1)) * (i % 5 == 0))
        theta = float(rng.uniform(-3.2, 3.2))
        n = 150_001
        ref = oracle.mc_poly_pair(robot, pos, theta, obstacle, sd, 9, i, 11, n)
        got = gpu_hits(eng, robot, pos, theta, obstacle, sd, 9, i, 11, n)
        assert got == ref, (i, ka, kb, pos, sd, got, ref)


def test_mc_poly_pair_sample_ranges_add_up(eng, oracle, wl):
    """disjoint sample ranges (what ranks would take) add up to the whole; ranges starting inside a group of four"""
    sc = wl.mc_poly_pair_scene()
    args = (sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1234, 0)
    whole = gpu_hits(eng, *args, 0, 1_000_000)
    assert whole == oracle.mc_poly_pair(*args, 0, 1_000_000)
    cuts = [0, 1, 333_333, 333_334, 700_001, 1_000_000]
    assert sum(gpu_hits(eng, *args, a, b - a) for a, b in zip(cuts, cuts[1:])) == whole
    d = eng.zeros(1, np.uint64)
    for a, b in zip(cuts, cuts[1:]):  # accumulating into one counter
        eng.mc_poly_pair(*args[:5], 1234, 0, a, b - a, d)
    assert int(d.get()[0]) == whole
    for begin, n in (((1 << 40) + 3, 20_047), ((1 << 32) - 5, 4_099), ((1 << 61) + 1, 777)):  # sample indices beyond 32 bits, ranges that straddle 2^32
        assert gpu_hits(eng, *args, begin, n) == oracle.mc_poly_pair(*args, begin, n), begin


def test_rectangles_as_polygons_reproduce_mc_pair(eng, oracle, wl):
    """The known-answer test of the extension: with sigma_w = sigma_h = 0 a rectangle given as a 4-gon gets, sample for sample, the
    very vertices c2d_mc_pair gives it.  For axis-aligned rectangles the hit counts are identical at 1e8 samples; for rotated ones
    the two tests differ in the scale of their axes (edge vector against normal), which moves a boolean only for a sample within
    an ulp of touching: a handful in 1e8 (4 for the config-3 scene — the oracles say the same, which this also checks)."""
    robot, obstacle = wl.rect_polygon(W, H), wl.rect_polygon(2.0, 1.0)
    n = 100_000_000
    for pos, theta, sd, seed in [((3.0, 1.0), 0.0, (0.3, 0.3, 0.0, 0, 0), 1), ((2.5, 0.2), 0.0, (0.5, 0.2, 0.0, 0, 0), 2)]:
        d = eng.zeros(1, np.uint64)
        eng.mc_pair(W, H, pos, (2.0, 1.0, theta), sd, seed, 0, 0, n, d)
        rect = int(d.get()[0])
        assert gpu_hits(eng, robot, pos, theta, obstacle, sd, seed, 0, 0, n) == rect
    sc = wl.MC_PAIR_SCENE
    d = eng.zeros(1, np.uint64)
    eng.mc_pair(W, H, sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, n, d)
    rect = int(d.get()[0])
    poly = gpu_hits(eng, robot, sc["pos"], sc["pose"][2], obstacle, sc["std_dev"], 1234, 0, 0, n)
    assert poly == oracle.mc_poly_pair(robot, sc["pos"], sc["pose"][2], obstacle, sc["std_dev"], 1234, 0, 0, n)
    assert abs(poly - rect) <= 20, (poly, rect)
    assert abs(poly - rect) / n < 1e-3  # BASELINE tolerance, by five orders of magnitude


def test_mc_poly_pair_closed_forms_1e8(eng, wl):
    """BASELINE tolerance (1e-3 at 1e8 samples) against closed forms: boxes given as polygons with position noise along one axis,
    and with RELATIVE width noise (a w x h box with sigma_w / w has the reference's distribution of widths, utils.cu:152-155)."""
    n = 100_000_000
    robot = wl.rect_polygon(W, H)
    w, h = 2.0, 1.0
    # an octagon-sampled box: extra vertices on the edges do not change the shape
    ox = np.array([-1, 0, 1, 1, 1, 0, -1, -1], np.float32) * np.float32(w / 2)
    oy = np.array([-1, -1, -1, 0, 1, 1, 1, 0], np.float32) * np.float32(h / 2)
    px, sx = 3.4, 0.5
    p = norm.cdf((px + (W + w) / 2) / sx) - norm.cdf((px - (W + w) / 2) / sx)
    got = gpu_hits(eng, robot, (px, 0.0), 0.0, (ox, oy), (sx, 0, 0, 0, 0), 7, 0, 0, n) / n
    assert abs(got - p) < 1e-3 and abs(got - p) < 5 * math.sqrt(p * (1 - p) / n) + 2e-5, (got, p)
    # width noise only: the obstacle's half width is (w / 2)(1 + s n), collision iff |px| < W / 2 + that
    px, srel = 3.3, 0.25
    z = ((px - W / 2) / (w / 2) - 1) / srel  # 1 + srel n > (px - W/2) / (w/2)   or   -(1 + srel n) > the same
    zneg = (-(px - W / 2) / (w / 2) - 1) / srel
    p = (1 - norm.cdf(z)) + norm.cdf(zneg)
    got = gpu_hits(eng, robot, (px, 0.0), 0.0, (ox, oy), (0, 0, 0, srel, 0), 8, 1, 0, n) / n
    assert abs(got - p) < 1e-3 and abs(got - p) < 5 * math.sqrt(p * (1 - p) / n) + 2e-5, (got, p)


def test_mc_poly_pair_1e8_vs_oracle_exact(eng, oracle, wl):
    """the bench scene of the polygon leg at 1e8 samples, hit for hit"""
    sc = wl.mc_poly_pair_scene()
    args = (sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1234, 0)
    n = 100_000_000
    assert gpu_hits(eng, *args, 0, n) == oracle.mc_poly_pair(*args, 0, n)


def test_mc_poly_pair_non_finite_scenes(eng, oracle, wl):
    """scenes with NaN / infinite / overflowing parameters: every sample in full, extremes as thrust::minmax_element has them"""
    sc = wl.mc_poly_pair_scene(6, 7)
    rx, ry = sc["robot"]
    ox, oy = sc["obstacle"]
    cases = []
    for bad in (np.nan, np.inf, -np.inf, 3e38, 2e15):
        r2 = rx.copy(); r2[0] = bad
        cases.append(((r2, ry), sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"]))
        r3 = ry.copy(); r3[3] = bad
        cases.append(((rx, r3), sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"]))
        o2 = ox.copy(); o2[0] = bad
        cases.append((sc["robot"], sc["pos"], sc["theta"], (o2, oy), sc["std_dev"]))
        o3 = oy.copy(); o3[5] = bad
        cases.append((sc["robot"], sc["pos"], sc["theta"], (ox, o3), sc["std_dev"]))
        cases.append((sc["robot"], (bad, 1.0), sc["theta"], sc["obstacle"], sc["std_dev"]))
        cases.append((sc["robot"], sc["pos"], bad, sc["obstacle"], sc["std_dev"]))
        for k in range(5):
            sd = list(sc["std_dev"]); sd[k] = bad
            cases.append((sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], tuple(sd)))
    with np.errstate(all="ignore"):
        for i, (robot, pos, theta, obstacle, sd) in enumerate(cases):
            ref = oracle.mc_poly_pair(robot, pos, theta, obstacle, sd, 5, i, 3, 20_001)
            got = gpu_hits(eng, robot, pos, theta, obstacle, sd, 5, i, 3, 20_001)
            assert got == ref, (i, got, ref)


def test_mc_poly_pair_large_finite_scenes(eng, oracle, wl):
    """finite parameters on both sides of the bounds within which the fast evaluation runs (lengths 1e-15 .. 1e8, relative shape deviations
    below 1e4): above them an intermediate can overflow, inf - inf makes a NaN, and only the all-bit-patterns path follows minmax_element there"""
    sc = wl.mc_poly_pair_scene(6, 7)
    rx, ry = sc["robot"]
    ox, oy = sc["obstacle"]
    cases = []
    for scale in (3e6, 9e7, 1.1e8, 4e9, 3e13):       # the whole scene at another scale (shape noise off and on)
        for sd in ((0.3 * scale, 0.3 * scale, 0.2, 0.0, 0.0), (0.3 * scale, 0.2 * scale, 0.2, 0.05, 0.1)):
            cases.append(((rx * scale, ry * scale), (sc["pos"][0] * scale, sc["pos"][1] * scale), sc["theta"], (ox * scale, oy * scale), sd))
    for sw in (3e3, 9.9e3, 1.1e4, 1e9, 5e14):        # huge relative shape deviations, scene at 1, 1e5 and 9e7
        for scale in (1.0, 1e5, 9e7):
            cases.append(((rx * scale, ry * scale), (sc["pos"][0] * scale, sc["pos"][1] * scale), sc["theta"], (ox * scale, oy * scale),
                          (0.3 * scale, 0.3 * scale, 0.2, sw, 0.5 * sw)))
    for e in (-14, -16, -19, -21, -22, -23, -26, -30):  # and whole scenes so small that products of two lengths are denormal: the pretest's margins
        scale = 10.0 ** e                                # are relative rounding bounds (scenes at 1e-22 and 1e-23 differed before lengths below 1e-15 left the fast path)
        for sd in ((0.3 * scale, 0.3 * scale, 0.2, 0.0, 0.0), (3.0 * scale, 3.0 * scale, 0.2, 0.05, 0.1)):
            cases.append(((rx * scale, ry * scale), (sc["pos"][0] * scale, sc["pos"][1] * scale), sc["theta"], (ox * scale, oy * scale), sd))
    with np.errstate(all="ignore"):
        for i, (robot, pos, theta, obstacle, sd) in enumerate(cases):
            robot = tuple(np.asarray(a, np.float32) for a in robot)
            obstacle = tuple(np.asarray(a, np.float32) for a in obstacle)
            ref = oracle.mc_poly_pair(robot, pos, theta, obstacle, sd, 9, i, 1, 30_001)
            got = gpu_hits(eng, robot, pos, theta, obstacle, sd, 9, i, 1, 30_001)
            assert got == ref, (i, got, ref)


def test_mc_poly_pair_argument_errors(eng, pkg, wl):
    sc = wl.mc_poly_pair_scene()
    d = eng.zeros(1, np.uint64)
    bad = pkg.make_polygon(*sc["robot"])
    bad.k = 17
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_pair(bad, sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1, 0, 0, 100, d)
    bad.k = 0
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_pair(sc["robot"], sc["pos"], sc["theta"], bad, sc["std_dev"], 1, 0, 0, 100, d)
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_pair(sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1, 0, 0, 100, None)
    eng.mc_poly_pair(sc["robot"], sc["pos"], sc["theta"], sc["obstacle"], sc["std_dev"], 1, 0, 0, 0, d)  # nothing to do
    assert int(d.get()[0]) == 0


# ---- many scenes, adaptive ------------------------------------------------------------------------------------------------------
def run_poly_scenes(eng, pkg, robot, poses, sds, scenes, max_samples, seed, base=0, schedule=(0, 0, 0), bins=(0, .01, .1, 1), acc=(1e-4, 1e-3, 1e-2)):
    n = len(scenes)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u, d_r = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32), eng.empty(n, pkg.ROW_DT)
    total, iters = eng.mc_poly_scenes(robot, d_p, len(poses), d_s, len(sds), d_sc, n, bins, acc, max_samples, seed, base, d_h, d_u, d_r, schedule=schedule)
    out = d_h.get(), d_u.get(), d_r.get(), total, iters
    for a in (d_p, d_s, d_sc, d_h, d_u, d_r):
        a.free()
    return out


def test_mc_poly_scenes_golden(eng, pkg):
    g = np.load(os.path.join(GOLD, "mc_poly_scenes_48.npz"))
    robot = (g["robot_x"], g["robot_y"])
    h, u, r, total, _ = run_poly_scenes(eng, pkg, robot, g["poly_poses"].view(pkg.POLY_POSE_DT).reshape(-1), g["std_devs"].view(pkg.STD_DT).reshape(-1),
                                        g["scenes"].view(pkg.SCENE_DT).reshape(-1), int(g["max_samples"]), int(g["seed"]), int(g["scene_id_base"]))
    assert np.array_equal(h, g["hits"]) and np.array_equal(u, g["n_used"])
    assert np.array_equal(r.view(np.uint32), g["rows"].view(np.uint32))
    assert total == int(g["total_samples"])


@pytest.mark.parametrize("n,max_samples,shape", [(1, 120_000, False), (63, 120_000, True), (3000, 120_000, False), (1500, 520_000, True)])
def test_mc_poly_scenes_matches_oracle(eng, oracle, pkg, wl, n, max_samples, shape):
    poses, sds = wl.random_poly_tables(97, 53, seed=n, shape_variance=shape)
    sc = wl.mc_poly_pair_scene(9, 5, seed=n + 1)
    scenes = wl.random_poly_scenes(n, poses, sds, 2.3, seed=n + 2)
    rh, ru, rr, rt = oracle.mc_poly_scenes(sc["robot"], poses, sds, scenes, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], max_samples, 21, 1000)
    h, u, r, total, iters = run_poly_scenes(eng, pkg, sc["robot"], poses, sds, scenes, max_samples, 21, 1000)
    assert np.array_equal(h, rh) and np.array_equal(u, ru)
    assert np.array_equal(r.view(np.uint32), rr.view(np.uint32))
    assert total == rt
    assert iters >= 1


def test_mc_poly_scenes_constant_schedule_and_other_bins(eng, oracle, pkg, wl):
    """ztest's constant schedule (ztest.cu:332-339), other accuracy bins, a fixed-samples run (max_samples = one batch)"""
    poses, sds = wl.random_poly_tables(40, 40, seed=4)
    sc = wl.mc_poly_pair_scene(5, 5, seed=8)
    scenes = wl.random_poly_scenes(700, poses, sds, 2.3, seed=9)
    for sched, bins, acc, ms in [((10000, 10000, 0), [0, .5, 1], [2e-3, 5e-3], 200_000), ((0, 0, 0), [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 1000),
                                 ((300, 5000, 1500), [0, .2, 1], [1e-3, 1e-2], 40_000)]:
        rh, ru, rr, rt = oracle.mc_poly_scenes(sc["robot"], poses, sds, scenes, bins, acc, ms, 2, 0, schedule=sched)
        h, u, r, total, _ = run_poly_scenes(eng, pkg, sc["robot"], poses, sds, scenes, ms, 2, 0, schedule=sched, bins=bins, acc=acc)
        assert np.array_equal(h, rh) and np.array_equal(u, ru) and total == rt
        assert np.array_equal(r.view(np.uint32), rr.view(np.uint32))


def test_mc_poly_scenes_bad_count_in_table_is_reported(eng, oracle, pkg, wl):
    """a vertex count outside 1..16 in the DEVICE table: clamped as the oracle clamps it, reported at the next synchronise"""
    poses, sds = wl.random_poly_tables(8, 8, seed=1)
    poses["obstacle"]["k"][3] = 40
    poses["obstacle"]["k"][5] = 0
    sc = wl.mc_poly_pair_scene(4, 4)
    scenes = wl.random_poly_scenes(64, poses, sds, 2.3, seed=2)
    scenes["pose_idx"][:16] = 3
    scenes["pose_idx"][16:32] = 5
    rh, ru, _, _ = oracle.mc_poly_scenes(sc["robot"], poses, sds, scenes, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 20_000, 6, 0)
    n = len(scenes)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32)
    eng.mc_poly_scenes(sc["robot"], d_p, len(poses), d_s, len(sds), d_sc, n, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 20_000, 6, 0, d_h, d_u, None, host_outputs=False)
    with pytest.raises(pkg.C2DError):
        eng.synchronize()
    eng.synchronize()  # reported once
    assert np.array_equal(d_h.get(), rh) and np.array_equal(d_u.get(), ru)


def test_mc_poly_scenes_argument_errors(eng, pkg, wl):
    poses, sds = wl.random_poly_tables(4, 4, seed=1)
    sc = wl.mc_poly_pair_scene()
    scenes = wl.random_poly_scenes(8, poses, sds, 2.3)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u = eng.zeros(8, np.uint32), eng.zeros(8, np.uint32)
    ok = dict(accuracy_bins=[0, .01, .1, 1], bin_accuracy=[1e-4, 1e-3, 1e-2], max_samples=2000, seed=1, scene_id_base=0, hits=d_h, n_used=d_u)
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_scenes(None, d_p, 4, d_s, 4, d_sc, 8, **ok)
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_scenes(sc["robot"], None, 4, d_s, 4, d_sc, 8, **ok)
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_scenes(sc["robot"], d_p, 0, d_s, 4, d_sc, 8, **ok)
    with pytest.raises(pkg.C2DError):
        eng.mc_poly_scenes(sc["robot"], d_p, 4, d_s, 4, d_sc, 8, **dict(ok, max_samples=0))
    assert eng.mc_poly_scenes(sc["robot"], d_p, 4, d_s, 4, d_sc, 0, **ok) == (0, 0)  # nothing to do
