"""CPU tests of the oracle itself: golden vectors, analytic known answers, and
agreement of the two independent restatements (numpy and C).  The reference has
no tests to mirror; these are the KATs that section derives."""
import math
import os

import numpy as np
import pytest
from scipy.stats import norm

from oracle import sat

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_rect_1k_numpy_and_c_agree(oracle):
    g = np.load(os.path.join(GOLD, "sat_rect_1k.npz"))
    planes, expected = g["planes"], g["expected"]
    got_c, cnt = oracle.sat_rect_pairs_verts(planes)
    got_np = sat.convex_collide(planes[:8].T, planes[8:].T)
    assert np.array_equal(got_c, expected)
    assert np.array_equal(got_np, expected)
    assert cnt == int(expected.sum())
    # pure-Python scalar form on the KAT block and a few random ones
    for i in list(range(int(g["n_kat"]))) + [100, 500, 999]:
        assert sat.convex_collide_scalar(planes[:8, i], planes[8:, i]) == expected[i]


def test_golden_rect_poses_rebuild_vertices(oracle):
    g = np.load(os.path.join(GOLD, "sat_rect_1k.npz"))
    k = int(g["n_kat"])
    poses, planes = g["poses"][:, k:], g["planes"][:, k:]
    r1 = oracle.rects_from_poses(*poses[:5])
    r2 = oracle.rects_from_poses(*poses[5:])
    assert np.array_equal(np.concatenate([r1, r2]).view(np.uint32), planes.view(np.uint32))
    out, _ = oracle.sat_rect_pairs_pose(poses)
    assert np.array_equal(out, g["expected"][k:])


def test_sat_known_answers(oracle):
    def rect(cx, cy, w, h):
        return np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy - h / 2, cx + w / 2, cy + h / 2, cx - w / 2, cy + h / 2], np.float32)

    a = rect(0, 0, 2, 2)
    assert oracle.convex_collide(a, a) == 1
    assert oracle.convex_collide(a, rect(5, 0, 2, 2)) == 0
    assert oracle.convex_collide(a, rect(2, 0, 2, 2)) == 1      # touching counts (strict <, utils.cu:178)
    assert oracle.convex_collide(a, rect(2, 2, 2, 2)) == 1      # corner touching
    assert oracle.convex_collide(a, rect(np.nextafter(np.float32(2), np.float32(3)), 0, 2, 2)) == 0
    diamond = np.array([1.75, 0.75, 2.75, 1.75, 1.75, 2.75, 0.75, 1.75], np.float32)
    assert oracle.convex_collide(a, diamond) == 0                # needs the rotated axes


def test_sat_symmetry_and_cyclic_shift(oracle, wl):
    poses = wl.random_obb_pose_planes(20000, seed=77, extent=4.0)
    r1 = oracle.rects_from_poses(*poses[:5])
    r2 = oracle.rects_from_poses(*poses[5:])
    ab, _ = oracle.sat_rect_pairs_verts(np.concatenate([r1, r2]))
    ba, _ = oracle.sat_rect_pairs_verts(np.concatenate([r2, r1]))
    assert np.array_equal(ab, ba)
    r1s = np.roll(r1, -2, axis=0)  # start the vertex list at vertex 1
    sh, _ = oracle.sat_rect_pairs_verts(np.concatenate([r1s, r2]))
    assert np.array_equal(ab, sh)
    # exactly representable joint translation
    t = np.float32(16.0)
    tr, _ = oracle.sat_rect_pairs_verts(np.concatenate([r1 + t, r2 + t]))
    assert (tr != ab).mean() < 1e-3  # translation changes rounding only on razor-edge pairs


def test_create_rect_order(oracle):
    r = oracle.create_rect(4.0, 2.0)
    assert r.tolist() == [-2, -1, 2, -1, 2, 1, -2, 1]      # utils.cu:122-129, CCW from (-,-)
    assert np.array_equal(sat.create_rect([4.0], [2.0])[0], r)


def test_canonical_math_accuracy(oracle):
    rng = np.random.default_rng(0)
    x = rng.uniform(-20, 20, 5000).astype(np.float32)
    s, c = oracle.sincosf(x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2e-7
    y = rng.integers(0, 2**32, 5000, dtype=np.uint64).astype(np.uint32)
    y[:4] = [0, 0x20000000, 0x40000000, 0xFFFFFFFF]
    s, c = oracle.sincos_u32(y)
    a = y.astype(np.float64) * (2 * np.pi / 2**32)
    assert np.abs(s - np.sin(a)).max() < 2e-7 and np.abs(c - np.cos(a)).max() < 2e-7
    u = np.concatenate([rng.uniform(2.0**-33, 1, 5000), [2.0**-33, 1.0, 0.5, 2 / 3]]).astype(np.float32)
    lg = oracle.logf(u)
    ref = np.log(u.astype(np.float64))
    assert np.abs(lg - ref).max() < 3e-6 and lg[-3] == 0.0
    assert np.all(lg <= 0)


def test_philox_known_answer(oracle):
    # Random123 kat_vectors: philox4x32-10, counter = key = 0 / all ones / pi digits
    assert oracle.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_stream_golden(oracle):
    g = np.load(os.path.join(GOLD, "philox_stream.npz"))
    raw = oracle.raw8(int(g["seed"]), int(g["scene"]), int(g["raw_begin"]), 16)
    words = oracle.draw_words(int(g["seed"]), int(g["scene"]), int(g["sample_begin"]), 16)
    nrm = oracle.normals5(int(g["seed"]), int(g["scene"]), int(g["sample_begin"]), 16)
    assert np.array_equal(raw, g["raw"])
    assert np.array_equal(words, g["draw_words"])
    assert np.array_equal(nrm.view(np.uint32), g["normals"].view(np.uint32))


def test_draw_layout_is_groups_of_four(oracle):
    """Sample s = 4g + j takes word j of blocks 8g and 8g+1 (first Box-Muller pair), words 2(j&1).. of block 8g+2+(j>>1)
    (second pair) and of block 8g+4+(j>>1) (third pair) — restated here straight from the Philox block function."""
    seed, scene = 0xDEADBEEFCAFEF00D, 0x123456789
    key = [seed & 0xffffffff, seed >> 32]
    for s0 in (0, 5, (1 << 40) + 3):
        words = oracle.draw_words(seed, scene, s0, 9)
        for i in range(9):
            s = s0 + i
            g, j = s >> 2, s & 3

            def block(b):
                blk = 8 * g + b
                return oracle.philox([blk & 0xffffffff, blk >> 32, scene & 0xffffffff, scene >> 32], key)

            o = 2 * (j & 1)
            want = [block(0)[j], block(1)[j], block(2 + (j >> 1))[o], block(2 + (j >> 1))[o + 1], block(4 + (j >> 1))[o], block(4 + (j >> 1))[o + 1]]
            assert words[i].tolist() == [int(v) for v in want]


def test_normals_are_standard_normal(oracle):
    n = oracle.normals5(42, 3, 0, 100000)
    assert np.abs(n.mean(0)).max() < 0.02 and np.abs(n.std(0) - 1).max() < 0.02
    assert np.abs(np.corrcoef(n.T) - np.eye(5)).max() < 0.02


def test_calc_slack_and_get_bin(oracle):
    # SURVEY.md §4.4
    for n in (1000, 20000, 4020000):
        assert oracle.calc_slack(n, 0) == pytest.approx(math.log(40.0) / n, rel=1e-6)
        assert oracle.calc_slack(n, n) == pytest.approx(math.log(40.0) / n, rel=1e-6)
    n, k = 20000, 5000
    p = k / n
    assert oracle.calc_slack(n, k) == pytest.approx(1.96 * math.sqrt(p * (1 - p) / n), rel=1e-5)
    # D1: the reference's int*int overflows beyond 46 340 hits; the fix stays finite and correct
    n, k = 4020000, 3000000
    p = k / n
    assert oracle.calc_slack(n, k) == pytest.approx(1.96 * math.sqrt(p * (1 - p) / n), rel=1e-3)
    bins = [0.0, 0.01, 0.1, 1.0]
    assert oracle.get_bin(0.0, bins) == 0
    assert oracle.get_bin(0.005, bins) == 0
    assert oracle.get_bin(np.float32(0.01), bins) == 1     # boundary goes to the higher bin (last match wins)
    assert oracle.get_bin(0.05, bins) == 1
    assert oracle.get_bin(np.float32(0.1), bins) == 2
    assert oracle.get_bin(1.0, bins) == 2                  # D2: no read past the end
    assert oracle.get_bin(2.0, bins) == 0                  # outside every bin -> 0, as the reference


def test_mc_closed_form_x_only(oracle):
    """robot theta=0 at (px, 0), only sigma_x > 0:
    p = Phi((px+a)/sx) - Phi((px-a)/sx), a = (W_robot + w)/2."""
    W, H, w, h, px, sx = 4.07, 1.74, 2.0, 1.0, 3.4, 0.5
    n = 400000
    hits = oracle.mc_pair(W, H, (px, 0.0), (w, h, 0.0), (sx, 0, 0, 0, 0), 7, 0, 0, n)
    a = (W + w) / 2
    p = norm.cdf((px + a) / sx) - norm.cdf((px - a) / sx)
    assert abs(hits / n - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4


def test_mc_closed_form_y_only(oracle):
    W, H, w, h, py, sy = 4.07, 1.74, 2.0, 1.0, 1.6, 0.4
    n = 400000
    hits = oracle.mc_pair(W, H, (0.0, py), (w, h, 0.0), (0, sy, 0, 0, 0), 8, 1, 0, n)
    a = (H + h) / 2
    p = norm.cdf((py + a) / sy) - norm.cdf((py - a) / sy)
    assert abs(hits / n - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4


def test_mc_closed_form_width_only(oracle):
    """Only sigma_w > 0 (shape variance): obstacle half width becomes w/2 + N(0, sw)/2;
    collision iff |px| <= W/2 + w/2 + dw/2 (robot left of / right of the box)."""
    W, H, w, h, px, sw = 4.07, 1.74, 2.0, 1.0, 3.3, 0.5
    n = 400000
    hits = oracle.mc_pair(W, H, (px, 0.0), (w, h, 0.0), (0, 0, 0, sw, 0), 9, 2, 0, n)
    # hw + dw/2 >= px - W/2  <=>  dw >= 2*(px - W/2 - w/2); for negative half width the
    # box is mirrored and still collides when |hw'| reaches the robot — negligible here
    t = 2 * (px - W / 2 - w / 2) / sw
    p = 1 - norm.cdf(t)
    assert abs(hits / n - p) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4


def test_mc_pair_golden_and_range_additivity(oracle):
    g = np.load(os.path.join(GOLD, "mc_pair_cases.npz"))
    W, H = (float(v) for v in g["robot"])
    prm = g["params"][0]
    args = (W, H, tuple(prm[0:2]), tuple(prm[2:5]), tuple(prm[5:10]), int(g["seed"]), int(g["scene_id"][0]))
    b = int(g["sample_begin"])
    whole = oracle.mc_pair(*args, b, 20000)
    parts = oracle.mc_pair(*args, b, 7001) + oracle.mc_pair(*args, b + 7001, 12999)
    assert whole == parts


def test_mc_scenes_golden_subset(oracle):
    g = np.load(os.path.join(GOLD, "mc_scenes_64.npz"))
    idx = np.where(g["n_used"] <= 9000)[0][:6]
    scenes = g["scenes"][idx]
    W, H = 4.07, 1.74
    # scene ids are positional: evaluate each selected scene with its own id
    for j, i in enumerate(idx):
        hits, used, rows, _ = oracle.mc_scenes(g["poses"], g["std_devs"], scenes[j:j + 1], W, H, [0, .01, .1, 1],
                                               [1e-4, 1e-3, 1e-2], int(g["max_samples"]), int(g["seed"]), int(i))
        assert hits[0] == g["hits"][i] and used[0] == g["n_used"][i]
        assert rows["cp"][0] == g["rows"]["cp"][i]


def test_poly_golden_and_rect_consistency(oracle):
    g = np.load(os.path.join(GOLD, "poly_k16_1k.npz"))
    out, _ = oracle.sat_poly_pairs(g["vx"], g["vy"], g["k"])
    assert np.array_equal(out, g["expected"])
    assert np.array_equal(sat.poly_collide_batch(g["vx"], g["vy"], g["k"]), g["expected"])
    # a rectangle pair fed through the polygon path (true normals) gives the same
    # answer as the rectangle path (edge vectors) away from razor-edge cases
    r = np.load(os.path.join(GOLD, "sat_rect_1k.npz"))
    planes, n = r["planes"], 1000
    vx = np.zeros((2, 16, n), np.float32)
    vy = np.zeros((2, 16, n), np.float32)
    for p in range(2):
        for v in range(4):
            vx[p, v] = planes[8 * p + 2 * v]
            vy[p, v] = planes[8 * p + 2 * v + 1]
    k = np.full((2, n), 4, np.uint8)
    out, _ = oracle.sat_poly_pairs(vx, vy, k)
    assert (out != r["expected"]).sum() <= 2


def test_poly_rejects_bad_vertex_count(oracle):
    vx = np.zeros((2, 16, 4), np.float32)
    k = np.array([[3, 3, 0, 3], [3, 3, 3, 3]], np.uint8)
    with pytest.raises(ValueError):
        oracle.sat_poly_pairs(vx, vx, k)


def test_sample_scenes_distribution(oracle, wl):
    poses, sds, _ = wl.random_tables(64, 64, seed=3)
    s = oracle.sample_scenes(poses, sds, 4.07, 1.74, 4.0, 5, 0, 4000)
    assert s["pose_idx"].min() >= 0 and s["pose_idx"].max() < 64
    assert s["var_idx"].min() >= 0 and s["var_idx"].max() < 64
    assert len(np.unique(s["pose_idx"])) > 50
    r = np.hypot(s["x"], s["y"])
    assert 2.0 < np.median(r) < 12.0


def test_hypothesis_numpy_and_c_oracles_agree_on_arbitrary_floats(oracle):
    """Property test over arbitrary finite vertex values (subnormals, zeros, large magnitudes):
    the numpy and C restatements agree, and the answer is symmetric in the two rectangles."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from hypothesis.extra.numpy import arrays

    f = st.floats(min_value=-float(2**50), max_value=float(2**50), allow_nan=False, allow_infinity=False, width=32)

    @settings(max_examples=300, deadline=None)
    @given(arrays(np.float32, (8,), elements=f), arrays(np.float32, (8,), elements=f))
    def check(r1, r2):
        c = oracle.convex_collide(r1, r2)
        assert c == sat.convex_collide_scalar(r1, r2)
        assert c == int(sat.convex_collide(r1[None, :], r2[None, :])[0])
        assert c == oracle.convex_collide(r2, r1)

    check()


def test_contraction_variants_of_the_oracle(oracle, wl):
    """oracle/libc2d_oracle_fmad{1,2}.so: the reference's dot products contracted as nvcc -fmad=true might
    (oracle/tools/fmad_study.py holds the full-size study).  Small-size facts: the golden config-1 booleans do not
    depend on the convention; pairs built ON the decision boundary do; Monte-Carlo counts move by a handful."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sat_rect_1k.npz"))
    planes = np.ascontiguousarray(g["planes"])
    ref, _ = oracle.sat_rect_pairs_verts(planes)
    poses = wl.random_obb_pose_planes(200_000, seed=0x5A7)
    pref, _ = oracle.sat_rect_pairs_pose(poses)
    sc = wl.MC_PAIR_SCENE
    href = oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 1_000_000)
    for k in (1, 2):
        v = oracle.load_variant(f"fmad{k}")
        assert v.lib().c2d_oracle_fmad_variant() == k and oracle.lib().c2d_oracle_fmad_variant() == 0
        out, _ = v.sat_rect_pairs_verts(planes)
        assert np.array_equal(out, ref) and np.array_equal(out, g["expected"])
        vp = np.concatenate([v.rects_from_poses(*poses[:5]), v.rects_from_poses(*poses[5:])])
        cp = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
        assert 0.02 < (vp.view(np.uint32) != cp.view(np.uint32)).mean() < 0.2    # the vertices DO depend on it ...
        pout, _ = v.sat_rect_pairs_pose(poses)
        assert (pout != pref).sum() <= 2                                           # ... random pairs' booleans (almost) never
        h = v.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 1_000_000)
        assert abs(h - href) <= 5


def test_non_finite_vertices_follow_minmax_element(oracle, wl):
    """thrust::minmax_element (utils.cu:176-177) is comparison based: a NaN projection at element 0 stays the
    extreme (the axis then never separates), a NaN at a later element is skipped.  Both restatements agree on
    50 000 pairs with NaN / inf / overflowing coordinates, and on hand-made cases of each kind."""
    poses = wl.random_obb_pose_planes(50_000, seed=1, extent=2.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    bad = wl.inject_non_finite(planes, seed=3)
    assert np.isnan(bad).any() and np.isinf(bad).any()
    got_c, _ = oracle.sat_rect_pairs_verts(bad)
    got_np = sat.convex_collide(bad[:8].T, bad[8:].T)
    assert np.array_equal(got_c, got_np)
    for i in np.flatnonzero(~np.isfinite(bad).all(axis=0))[:40]:
        assert sat.convex_collide_scalar(bad[:8, i], bad[8:, i]) == got_c[i]

    def rect(cx, cy, w, h):
        return np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy - h / 2, cx + w / 2, cy + h / 2, cx - w / 2, cy + h / 2], np.float32)

    a, far = rect(0, 0, 2, 2), rect(10, 0, 2, 2)
    assert oracle.convex_collide(a, far) == 0
    nan0 = far.copy()
    nan0[0] = np.nan     # vertex 0 of rectangle 2: every axis sees a NaN first projection -> nothing separates
    assert oracle.convex_collide(a, nan0) == 1 and sat.convex_collide_scalar(a, nan0) == 1
    nan2 = far.copy()
    nan2[4] = np.nan     # vertex 2: its projections are skipped; rectangle 1's own axes still separate the rest
    assert oracle.convex_collide(a, nan2) == 0 and sat.convex_collide_scalar(a, nan2) == 0
    inf1 = far.copy()
    inf1[2] = np.inf     # an infinite vertex is ordered like any number
    assert oracle.convex_collide(a, inf1) == sat.convex_collide_scalar(a, inf1)


def test_non_finite_polygons_numpy_and_c_agree(oracle, wl):
    vx, vy, k = wl.random_convex_polygons(20_000, seed=8, extent=2.0)
    bx = wl.inject_non_finite(vx.reshape(32, -1), seed=4).reshape(vx.shape)
    by = wl.inject_non_finite(vy.reshape(32, -1), seed=5, frac=0.2).reshape(vy.shape)
    got_c, _ = oracle.sat_poly_pairs(bx, by, k)
    got_np = sat.poly_collide_batch(bx, by, k)
    assert np.array_equal(got_c, got_np)
    touched = np.flatnonzero(~(np.isfinite(bx).all(axis=(0, 1)) & np.isfinite(by).all(axis=(0, 1))))
    for i in touched[:25]:
        assert sat.poly_collide(bx[0, :, i], by[0, :, i], int(k[0, i]), bx[1, :, i], by[1, :, i], int(k[1, i])) == got_c[i]


def test_sat_agrees_with_exact_rational_arithmetic_away_from_the_boundary(oracle, wl):
    """An anchor outside this repository's own arithmetic: the separating-axis predicate of utils.cu:159-184 evaluated in exact
    rational arithmetic (fractions) on the same float32 vertices.  Wherever every axis' exact gap or overlap exceeds the float32
    rounding of its projections, the float32 oracle must give the same boolean — on random pairs that is all but a handful."""
    from fractions import Fraction

    poses = wl.random_obb_pose_planes(3000, seed=123, extent=3.0)
    r1 = oracle.rects_from_poses(*poses[:5])
    r2 = oracle.rects_from_poses(*poses[5:])
    got, _ = oracle.sat_rect_pairs_verts(np.concatenate([r1, r2]))
    decided = 0
    for i in range(r1.shape[1]):
        a = [Fraction(float(v)) for v in r1[:, i]]
        b = [Fraction(float(v)) for v in r2[:, i]]
        exact_collide, safe = True, True
        for r in (a, b):
            for e in range(4):
                nx = r[(e + 1) * 2 % 8] - r[e * 2]
                ny = r[((e + 1) * 2 + 1) % 8] - r[e * 2 + 1]
                p1 = [nx * a[2 * k] + ny * a[2 * k + 1] for k in range(4)]
                p2 = [nx * b[2 * k] + ny * b[2 * k + 1] for k in range(4)]
                gap = max(min(p2) - max(p1), min(p1) - max(p2))          # > 0: separated on this axis
                scale = max(abs(v) for v in p1 + p2) + Fraction(1, 10**30)
                if abs(gap) < scale * Fraction(1, 2**18):                  # within ~64 ulp of the decision: rounding may decide
                    safe = False
                if gap > 0:
                    exact_collide = False
        if safe:
            decided += 1
            assert int(exact_collide) == got[i], i
    assert decided > 2900


def test_parallel_axis_certificate_error_bound(oracle, wl):
    """The bound behind the kernels' certificates (c2d_math.hpp rect_collide_certified, c2d_mc.hip sample_collides_mask): for a
    rectangle's edge axes a (edge i) and b (edge i + 2), d = a + b, every float32 projection satisfies
    |p_b(v) + p_a(v)| <= |d|_1 C (1 + 3u) + 4u (1 + u) |a|_1 C with C >= every |coordinate|.  Checked in float64 on 300 000
    rectangle pairs at several scales; the largest observed ratio is close to one (the d-term is attained), which is why the
    kernels' thresholds are the bound itself plus 2^-8, not a fraction of it."""
    F, u, worst = np.float32, 2.0 ** -24, 0.0
    for seed, extent, scale in ((1, 8.0, 1.0), (2, 1.0, 1.0), (3, 100.0, 1.0), (4, 8.0, 1e-3), (5, 8.0, 1e4), (6, 0.01, 1.0)):
        poses = wl.random_obb_pose_planes(50_000, seed=seed, extent=extent)
        r1 = (oracle.rects_from_poses(*poses[:5]) * F(scale)).astype(F)
        r2 = (oracle.rects_from_poses(*poses[5:]) * F(scale)).astype(F)
        C = np.maximum(np.abs(r1).max(0), np.abs(r2).max(0)).astype(np.float64)
        for r in (r1, r2):
            for i in range(2):
                ax, ay = (r[2 * i + 2] - r[2 * i]).astype(F), (r[2 * i + 3] - r[2 * i + 1]).astype(F)
                bx, by = (r[(2 * i + 6) & 7] - r[2 * i + 4]).astype(F), (r[(2 * i + 7) & 7] - r[2 * i + 5]).astype(F)
                d1 = np.abs(ax.astype(np.float64) + bx) + np.abs(ay.astype(np.float64) + by)
                a1 = np.abs(ax.astype(np.float64)) + np.abs(ay.astype(np.float64))
                bound = d1 * C * (1 + 3 * u) + 4 * u * (1 + u) * a1 * C
                for rr in (r1, r2):
                    for k in range(4):
                        pa = ((ax * rr[2 * k]).astype(F) + (ay * rr[2 * k + 1]).astype(F)).astype(F)
                        pb = ((bx * rr[2 * k]).astype(F) + (by * rr[2 * k + 1]).astype(F)).astype(F)
                        e = np.abs(pa.astype(np.float64) + pb.astype(np.float64))
                        assert (e <= bound).all(), (seed, i, k)
                        worst = max(worst, float(np.where(bound > 0, e / np.where(bound > 0, bound, 1), 0).max()))
    assert 0.5 < worst <= 1.0


# ---- the closed-form test behind the Monte-Carlo full evaluation and the pose-format pair kernel (c2d_mc.hip model_gap,
# c2d_sat.hip): a numpy restatement, decision by decision against the oracle's convex_collide
def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def closed_form_gap(oracle, x1, y1, w1, h1, t1, x2, y2, w2, h2, t2):
    """-> (gap, margin): float32 restatement of model_gap for rectangle 1 = 'robot', rectangle 2 = 'obstacle', with the margin
    (64 + 20 C / h_min) u C taken per pair"""
    F = np.float32
    s1, c1 = oracle.sincosf(t1)
    s2, c2 = oracle.sincosf(t2)
    hw, hh, hx, hy = np.abs(w1 / F(2)), np.abs(h1 / F(2)), np.abs(w2 / F(2)), np.abs(h2 / F(2))
    q1 = _fma32(c1, c2, (s1 * s2).astype(F))
    q2 = _fma32(c1, s2, -(s1 * c2).astype(F))
    ex, ey = (x2 - x1).astype(F), (y2 - y1).astype(F)
    t1_ = _fma32(c1, ex, (s1 * ey).astype(F))
    t2_ = _fma32(c1, ey, -(s1 * ex).astype(F))
    t3_ = _fma32(c2, ex, (s2 * ey).astype(F))
    t4_ = _fma32(c2, ey, -(s2 * ex).astype(F))
    a1, a2 = np.abs(q1), np.abs(q2)
    g1 = (np.abs(t1_) - _fma32(hx, a1, _fma32(hy, a2, hw))).astype(F)
    g2 = (np.abs(t2_) - _fma32(hx, a2, _fma32(hy, a1, hh))).astype(F)
    g3 = (np.abs(t3_) - _fma32(hw, a1, _fma32(hh, a2, hx))).astype(F)
    g4 = (np.abs(t4_) - _fma32(hw, a2, _fma32(hh, a1, hy))).astype(F)
    g = np.maximum(np.maximum(g1, g2), np.maximum(g3, g4))
    C = (np.maximum(np.maximum(np.abs(x1), np.abs(y1)) + (hw + hh), np.maximum(np.abs(x2), np.abs(y2)) + (hx + hy)) * F(1 + 2.0 ** -10)).astype(F)
    h = np.minimum(np.minimum(hw, hh), np.minimum(hx, hy))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        margin = ((F(64) + F(20) * (C / h)) * (F(2.0 ** -24) * C) * F(1 + 2.0 ** -10)).astype(F)
    margin = np.where((h >= F(1e-12)) & (C < F(1e15)), margin, np.float32(np.inf))
    return g, margin


def _closed_form_against_oracle(oracle, poses):
    F = np.float32
    poses = [np.asarray(p, F) for p in poses]
    g, m = closed_form_gap(oracle, *poses)
    ref, _ = oracle.sat_rect_pairs_pose(np.stack(poses))
    sep, col = g > m, g < -m
    assert not (sep & (ref != 0)).any(), "closed form says separated, the oracle says the pair collides"
    assert not (col & (ref == 0)).any(), "closed form says colliding, the oracle finds a separating axis"
    return float(1.0 - (sep | col).mean())


def test_closed_form_gap_decides_like_the_oracle_on_random_pairs(oracle):
    rng = np.random.default_rng(11)
    n = 60_000
    for pos, lo, hi, scale in ((8, 0.1, 5, 1), (8, 0.1, 5, 1e-3), (8, 0.1, 5, 1e4), (100, 0.1, 5, 1), (1, 0.5, 2, 1), (1000, 0.01, 1, 1), (3, 3, 6, 1), (2, 1e-3, 4, 1)):
        xy = [rng.uniform(-pos, pos, n) * scale for _ in range(4)]
        wh = [rng.uniform(lo, hi, n) * scale for _ in range(4)]
        th = [rng.uniform(-7, 7, n) for _ in range(2)]
        thin = _closed_form_against_oracle(oracle, (xy[0], xy[1], wh[0], wh[1], th[0], xy[2], xy[3], wh[2], wh[3], th[1]))
        assert thin < (0.2 if pos >= 100 or lo < 0.01 else 0.02), (pos, lo, hi, scale, thin)


def test_closed_form_gap_on_razor_thin_pairs(oracle, wl):
    """Pairs built to touch (workloads.touching_pose_pairs): every decision the closed form takes must be the oracle's."""
    for seed, (scale, off) in enumerate(((1.0, 0.0), (1.0, 50.0), (1e-4, 0.0), (1e3, 0.0))):
        thin = _closed_form_against_oracle(oracle, wl.touching_pose_pairs(40_000, seed=12 + seed, scale=scale, offset=off))
        assert thin > 0.01, "the construction is meant to land inside the margin often"


def test_closed_form_error_budget(oracle, wl):
    """The inequality the closed-form margin rests on (c2d_mc.hip model_gap): on every one of the reference's eight axes the smaller
    of the two computed overlaps of utils.cu:178 differs from -G(n) = -2 h G_e of the real model rectangles by at most 2E,
    E = 5.1u |a^|_1 C + 16.1u C^2.  Measured in float64 on random and on touching pairs; the worst ratio seen is about 0.15."""
    F, D, u = np.float32, np.float64, 2.0 ** -24
    rng = np.random.default_rng(5)
    worst = 0.0
    cases = []
    for pos, lo, hi, scale in ((8, 0.1, 5, 1), (8, 0.1, 5, 1e-3), (100, 0.1, 5, 1), (1000, 0.01, 1, 1), (3, 3, 6, 1e4)):
        n = 40_000
        xy = [rng.uniform(-pos, pos, n) * scale for _ in range(4)]
        wh = [rng.uniform(lo, hi, n) * scale for _ in range(4)]
        th = [rng.uniform(-7, 7, n) for _ in range(2)]
        cases.append(np.stack([xy[0], xy[1], wh[0], wh[1], th[0], xy[2], xy[3], wh[2], wh[3], th[1]]).astype(F))
    cases.append(wl.touching_pose_pairs(40_000, seed=3))
    cases.append(wl.touching_pose_pairs(40_000, seed=4, offset=30.0))
    for p in cases:
        x1, y1, w1, h1, t1, x2, y2, w2, h2, t2 = p
        r1, r2 = oracle.rects_from_poses(x1, y1, w1, h1, t1), oracle.rects_from_poses(x2, y2, w2, h2, t2)
        (s1, c1), (s2, c2) = oracle.sincosf(t1), oracle.sincosf(t2)
        hx1, hy1, hx2, hy2 = [np.abs(v.astype(D)) / 2 for v in (w1, h1, w2, h2)]
        C = np.maximum(np.maximum(np.abs(x1), np.abs(y1)).astype(D) + hx1 + hy1, np.maximum(np.abs(x2), np.abs(y2)).astype(D) + hx2 + hy2)
        for r, c, s, hx, hy in ((r1, c1, s1, hx1, hy1), (r2, c2, s2, hx2, hy2)):
            for i in range(4):
                ax, ay = (r[(2 * i + 2) & 7] - r[2 * i]).astype(F), (r[(2 * i + 3) & 7] - r[2 * i + 1]).astype(F)

                def interval(rr):
                    q = np.stack([((ax * rr[2 * k]).astype(F) + (ay * rr[2 * k + 1]).astype(F)).astype(F) for k in range(4)])
                    return q.min(0).astype(D), q.max(0).astype(D)

                (mn1, mx1), (mn2, mx2) = interval(r1), interval(r2)
                overlap = np.minimum(mx1 - mn2, mx2 - mn1)
                ex, ey, h = (c.astype(D), s.astype(D), hx) if i % 2 == 0 else (-s.astype(D), c.astype(D), hy)

                def extent(cc, ss, a, b):
                    cc, ss = cc.astype(D), ss.astype(D)
                    return a * np.abs(ex * cc + ey * ss) + b * np.abs(-ex * ss + ey * cc)

                gap = np.abs(ex * (x2.astype(D) - x1) + ey * (y2.astype(D) - y1)) - extent(c1, s1, hx1, hy1) - extent(c2, s2, hx2, hy2)
                E = 5.1 * u * (np.abs(ax.astype(D)) + np.abs(ay.astype(D))) * C + 16.1 * u * C * C
                ratio = np.abs(overlap + 2 * h * gap) / (2 * E)
                assert (ratio <= 1.0).all(), (i, float(ratio.max()))
                worst = max(worst, float(ratio.max()))
    assert 0.01 < worst < 0.5, worst


def test_sincos_is_a_rotation_to_within_2_pow_minus_20(oracle):
    """model_gap's budget writes a rectangle's extent along its own edge as h, not h (c^2 + s^2), and allows 16uC for it:
    sincos_ returns |c^2 + s^2 - 1| <= 2^-20 and |c|, |s| <= 1 for every finite angle tried (dense near the octant seams too)."""
    rng = np.random.default_rng(9)
    k = rng.integers(-40, 41, 60_000)
    seams = (k * (np.pi / 4) + rng.normal(0, 1e-6, k.size)).astype(np.float32)
    x = np.concatenate([rng.uniform(-7, 7, 120_000), rng.uniform(-1e5, 1e5, 30_000), rng.normal(0, 1e-3, 30_000), seams]).astype(np.float32)
    s, c = oracle.sincosf(x)
    s, c = s.astype(np.float64), c.astype(np.float64)
    assert np.abs(c * c + s * s - 1).max() <= 2.0 ** -20
    assert np.abs(c).max() <= 1.0 and np.abs(s).max() <= 1.0
