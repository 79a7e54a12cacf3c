"""GPU parity tests of the random stream and the Monte-Carlo kernels through the
C-ABI: raw Philox words, normals, hit counts and adaptive-stop results must be
bit-identical to the CPU oracle; probabilities are checked against closed forms
at 1e8 samples with the tolerance BASELINE.json states (1e-3)."""
import math
import os

import numpy as np
import pytest
from scipy.stats import norm

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
W, H = 4.07, 1.74


def gpu_hits(eng, pos, pose, sd, seed, scene, begin, n, robot=(W, H)):
    d = eng.zeros(1, np.uint64)
    eng.mc_pair(robot[0], robot[1], pos, pose, sd, seed, scene, begin, n, d)
    h = int(d.get()[0])
    d.free()
    return h


def test_philox_stream_golden_and_oracle(eng, oracle):
    g = np.load(os.path.join(GOLD, "philox_stream.npz"))
    seed, scene, begin = int(g["seed"]), int(g["scene"]), int(g["sample_begin"])
    d_n, d_r = eng.empty((16, 5), np.float32), eng.empty((16, 6), np.uint32)
    eng.philox_normals(seed, scene, begin, 16, d_n, d_r)
    assert np.array_equal(d_r.get(), g["draw_words"])
    assert np.array_equal(d_n.get().view(np.uint32), g["normals"].view(np.uint32))
    # a longer run against the live oracle, bit for bit
    n = 50000
    d_n, d_r = eng.empty((n, 5), np.float32), eng.empty((n, 6), np.uint32)
    eng.philox_normals(99, 12345, 10**12 + 1, n, d_n, d_r)  # starts inside a group of four
    assert np.array_equal(d_r.get(), oracle.draw_words(99, 12345, 10**12 + 1, n))
    assert np.array_equal(d_n.get().view(np.uint32), oracle.normals5(99, 12345, 10**12 + 1, n).view(np.uint32))


def _math_both(eng, oracle, fn, bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    d_in = eng.to_device(bits)
    d0, d1 = eng.empty(bits.size, np.float32), eng.empty(bits.size, np.float32)
    eng.math_eval(fn, d_in, bits.size, d0, d1)
    g0, g1 = d0.get(), d1.get()
    for a in (d_in, d0, d1):
        a.free()
    r0, r1 = oracle.math_eval(fn, bits)
    return g0.view(np.uint32), g1.view(np.uint32), r0.view(np.uint32), r1.view(np.uint32)


def test_canonical_math_bit_exact(eng, oracle):
    """Device canonical math == oracle, bit for bit, on dense input sweeps."""
    rng = np.random.default_rng(3)
    # sqrt: exhaustive over two binades [1, 4) (every mantissa, both exponent parities) + the Box-Muller range
    bits = np.arange(0x3F800000, 0x40800000, dtype=np.uint32)
    g0, _, r0, _ = _math_both(eng, oracle, eng.MATH_SQRT, bits)
    assert np.array_equal(g0, r0)
    x = np.concatenate([rng.uniform(1e-7, 50, 4_000_000), 10.0 ** rng.uniform(-25, 25, 1_000_000), [0.0, -0.0, 2.0**-96, 2.0**96]]).astype(np.float32)
    g0, _, r0, _ = _math_both(eng, oracle, eng.MATH_SQRT, x.view(np.uint32))
    assert np.array_equal(g0, r0)
    # log on (0, 1] (the Box-Muller domain) and beyond
    u = np.concatenate([rng.uniform(2.0**-33, 1, 4_000_000), 2.0 ** rng.uniform(-33, 0, 1_000_000), [2.0**-33, 1.0, 0.5, 2 / 3]]).astype(np.float32)
    g0, _, r0, _ = _math_both(eng, oracle, eng.MATH_LOG, u.view(np.uint32))
    assert np.array_equal(g0, r0)
    # sin/cos of float angles incl. large ones, and of integer angles incl. octant boundaries
    th = np.concatenate([rng.uniform(-10, 10, 4_000_000), rng.uniform(-1e5, 1e5, 1_000_000), rng.uniform(-1e9, 1e9, 500_000),
                         10.0 ** rng.uniform(-40, 15, 500_000), [0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 1e15, -1e15]]).astype(np.float32)
    g0, g1, r0, r1 = _math_both(eng, oracle, eng.MATH_SINCOS, th.view(np.uint32))
    assert np.array_equal(g0, r0) and np.array_equal(g1, r1)
    y = np.concatenate([rng.integers(0, 2**32, 5_000_000, dtype=np.uint64).astype(np.uint32),
                        np.array([0, 0x1FFFFFFF, 0x20000000, 0x20000001, 0x3FFFFFFF, 0x40000000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF], np.uint32)])
    g0, g1, r0, r1 = _math_both(eng, oracle, eng.MATH_SINCOS_U32, y)
    assert np.array_equal(g0, r0) and np.array_equal(g1, r1)
    g0, g1, r0, r1 = _math_both(eng, oracle, eng.MATH_BOX_MULLER, y)
    assert np.array_equal(g0, r0) and np.array_equal(g1, r1)


def test_mc_pair_golden_cases(eng):
    g = np.load(os.path.join(GOLD, "mc_pair_cases.npz"))
    rw, rh = (float(v) for v in g["robot"])
    for prm, sid, want in zip(g["params"], g["scene_id"], g["hits"]):
        got = gpu_hits(eng, tuple(prm[0:2]), tuple(prm[2:5]), tuple(prm[5:10]), int(g["seed"]), int(sid),
                       int(g["sample_begin"]), int(g["n_samples"]), (rw, rh))
        assert got == int(want)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097, 100000])
def test_mc_pair_sizes_vs_oracle(eng, oracle, wl, n):
    sc = wl.MC_PAIR_SCENE
    got = gpu_hits(eng, sc["pos"], sc["pose"], sc["std_dev"], 1234, 5, 7, n)
    assert got == oracle.mc_pair(W, H, sc["pos"], sc["pose"], sc["std_dev"], 1234, 5, 7, n)


def test_mc_pair_shape_variance_and_zero_sigma(eng, oracle):
    pos, pose = (2.9, -1.2), (1.5, 2.5, 1.1)
    for sd in [(0.3, 0.2, 0.1, 0.4, 0.5), (0.0, 0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 0.3, 0.0, 0.2), (0.2, 0.2, 0.0, 0.3, 0.0)]:
        got = gpu_hits(eng, pos, pose, sd, 77, 1, 0, 50000)
        assert got == oracle.mc_pair(W, H, pos, pose, sd, 77, 1, 0, 50000), sd


def test_mc_centre_pretest_boundary_sweep(eng, oracle):
    """The kernels skip a wave's samples when the obstacle centres alone prove a miss (bounding-disk
    argument in make_scene).  Sweep the robot through the region where that shortcut starts to fire —
    from clearly colliding to clearly apart, several orientations, with and without shape variance —
    and require the exact hit count of the oracle, which has no shortcut."""
    rng = np.random.default_rng(8)
    n = 150_000
    cases = 0
    for sd in [(0.3, 0.3, 0.2, 0.0, 0.0), (0.05, 0.4, 0.6, 0.0, 0.0), (0.2, 0.1, 0.1, 0.3, 0.2), (1e-3, 1e-3, 0.5, 0.0, 0.0)]:
        for pose_theta in (0.0, 0.6, 1.3):
            pose = (2.0, 1.0, pose_theta)
            rho = float(np.hypot(1.0 + 3.385 * sd[3], 0.5 + 3.385 * sd[4]))
            for dist in list(np.linspace(2.0, 2.035 + rho + 3.0, 9)) + [2.035 + rho * (1 + 2.0**-10) + k * 1e-4 for k in (-2, 0, 2)]:
                ang = float(rng.uniform(-0.4, 0.4))
                pos = (float(dist * np.cos(ang)), float(dist * np.sin(ang)))
                sid = cases
                got = gpu_hits(eng, pos, pose, sd, 2024, sid, 0, n)
                assert got == oracle.mc_pair(W, H, pos, pose, sd, 2024, sid, 0, n), (sd, pose, pos)
                cases += 1
    assert cases == 4 * 3 * 12


def test_mc_far_and_near_paths_unaligned_ranges(eng, oracle):
    """The kernels run a scene in one of two ways — groups of four samples per lane (near), or compacted candidates of the
    radius test (far; csrc/c2d_mc.hip) — chosen from the scene's candidate fraction.  Distances from overlapping to far
    apart cross that switch; sample ranges start and end anywhere inside a group of four."""
    sd = (0.3, 0.25, 0.2, 0.0, 0.1)
    pose = (2.0, 1.0, 0.4)
    k = 0
    for dist in (0.5, 2.0, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0, 7.0, 9.0, 14.0):
        for begin, n in ((1, 70_001), (4 * 12345 + 2, 33_333), ((1 << 40) + 3, 2_047), (7, 3), (6, 257)):
            pos = (dist * 0.8, dist * 0.6)
            got = gpu_hits(eng, pos, pose, sd, 31337, 100 + k, begin, n)
            assert got == oracle.mc_pair(W, H, pos, pose, sd, 31337, 100 + k, begin, n), (dist, begin, n)
            k += 1


@pytest.mark.parametrize("robot,pos,pose,sd", [
    ((4.07, 1.74), (0.0, 0.0), (2.0, 1.0, 0.3), (0.0, 0.0, 0.0, 0.0, 0.0)),        # no noise at all, overlapping: p = 1
    ((4.07, 1.74), (9.0, 0.0), (2.0, 1.0, 0.3), (0.0, 0.0, 0.0, 0.0, 0.0)),        # no noise, apart: p = 0
    ((4.07, 1.74), (3.0, 0.5), (0.0, 0.0, 0.0), (0.3, 0.3, 0.2, 0.0, 0.0)),        # point obstacle
    ((4.07, 1.74), (3.0, 0.5), (0.0, 3.0, 1.0), (0.3, 0.3, 0.8, 0.0, 0.0)),        # segment obstacle
    ((0.0, 0.0), (0.5, 0.2), (2.0, 1.0, 0.0), (0.3, 0.3, 0.2, 0.0, 0.0)),          # point robot
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (10.0, 10.0, 3.0, 0.0, 0.0)),      # huge noise
    ((4.07, 1.74), (3.0, 1.0), (0.5, 0.5, 0.6), (0.3, 0.3, 0.2, 2.0, 2.0)),        # shape noise >> size: negative widths happen
    ((4.07, 1.74), (1e4, -1e4), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),       # robot very far: large coordinates
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (1e-6, 1e-6, 1e-6, 0.0, 0.0)),     # almost no noise near the boundary
    ((4.07, 1.74), (2.9, 0.0), (2.0, 1.0, 0.0), (0.0, 0.3, 0.0, 0.0, 0.0)),        # sigma_x = 0 only
])
def test_mc_pair_degenerate_scenes(eng, oracle, robot, pos, pose, sd):
    """Degenerate sizes and noise levels: the pretest thresholds (rho = 0, G = 0, huge margins) must never
    change a result."""
    n = 60_000
    d = eng.zeros(1, np.uint64)
    eng.mc_pair(robot[0], robot[1], pos, pose, sd, 5, 9, 0, n, d)
    assert int(d.get()[0]) == oracle.mc_pair(robot[0], robot[1], pos, pose, sd, 5, 9, 0, n)
    d.free()


def test_mc_pair_range_additivity_and_accumulation(eng, wl):
    """Disjoint sample ranges sum to the whole (this is what sharding over GPUs relies on),
    and d_hits accumulates across calls."""
    sc = wl.MC_PAIR_SCENE
    args = (sc["pos"], sc["pose"], sc["std_dev"], 1234, 0)
    whole = gpu_hits(eng, *args, 0, 3_000_000)
    d = eng.zeros(1, np.uint64)
    for b, c in [(0, 1_000_001), (1_000_001, 999_999), (2_000_000, 1_000_000)]:
        eng.mc_pair(W, H, *args, b, c, d)
    assert int(d.get()[0]) == whole


def test_mc_pair_1e8_closed_form(eng):
    """BASELINE: MC probability within 1e-3 at 1e8 samples.  Closed forms of SURVEY.md §4.3."""
    n = 100_000_000
    px, sx, w, h = 3.4, 0.5, 2.0, 1.0
    p = norm.cdf((px + (W + w) / 2) / sx) - norm.cdf((px - (W + w) / 2) / sx)
    got = gpu_hits(eng, (px, 0.0), (w, h, 0.0), (sx, 0, 0, 0, 0), 7, 0, 0, n) / n
    assert abs(got - p) < 1e-3, (got, p)
    assert abs(got - p) < 5 * math.sqrt(p * (1 - p) / n) + 2e-5, (got, p)
    py, sy = 1.6, 0.4
    p = norm.cdf((py + (H + h) / 2) / sy) - norm.cdf((py - (H + h) / 2) / sy)
    got = gpu_hits(eng, (0.0, py), (w, h, 0.0), (0, sy, 0, 0, 0), 8, 1, 0, n) / n
    assert abs(got - p) < 1e-3 and abs(got - p) < 5 * math.sqrt(p * (1 - p) / n) + 2e-5, (got, p)


def test_mc_pair_1e8_vs_oracle_exact(eng, oracle, wl):
    """Config 3 at its stated size, hit for hit: the GPU's count over all 1e8 samples of the bench scene equals the oracle's
    (the loop of ccp.cu:135-139 over utils.cu:144-184; the OpenMP oracle walks 1e8 samples in a second or two), and so do the
    counts of an unaligned split of the same range (what two ranks would add up)."""
    sc = wl.MC_PAIR_SCENE
    args = (sc["pos"], sc["pose"], sc["std_dev"], int(os.environ.get("C2D_FULLSIZE_SEED", "1234"), 0), 0)   # (another stream: profiles/r06_fullsize_seeds.sh)
    n = 100_000_000
    ref = oracle.mc_pair(W, H, *args, 0, n)
    got = gpu_hits(eng, *args, 0, n)
    assert got == ref, (got, ref)
    cut = 37_000_001
    assert gpu_hits(eng, *args, 0, cut) + gpu_hits(eng, *args, cut, n - cut) == ref


def test_sample_scenes_matches_oracle(eng, oracle, wl, pkg):
    poses, sds, _ = wl.random_tables(500, 300, seed=21, shape_variance=True)
    n = 100_003
    ref = oracle.sample_scenes(poses, sds, W, H, 4.0, 99, 1000, n)
    d_p, d_s = eng.to_device(poses), eng.to_device(sds)
    d_sc = eng.empty(n, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 500, d_s, 300, W, H, 4.0, 99, 1000, n, d_sc)
    got = d_sc.get()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def run_scenes(eng, pkg, poses, sds, scenes, max_samples, seed, base=0):
    n = len(scenes)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u, d_r = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32), eng.empty(n, pkg.ROW_DT)
    total, iters = eng.mc_scenes(d_p, len(poses), d_s, len(sds), d_sc, n, W, H, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2],
                                 max_samples, seed, base, d_h, d_u, d_r)
    return d_h.get(), d_u.get(), d_r.get(), total, iters


def test_mc_scenes_golden_64(eng, pkg):
    g = np.load(os.path.join(GOLD, "mc_scenes_64.npz"))
    hits, used, rows, total, iters = run_scenes(eng, pkg, g["poses"], g["std_devs"], g["scenes"], int(g["max_samples"]), int(g["seed"]))
    assert np.array_equal(used, g["n_used"])
    assert np.array_equal(hits, g["hits"])
    assert np.array_equal(rows.view(np.uint32), g["rows"].view(np.uint32))
    assert total == int(g["total"])
    assert iters == 21  # 20 x 1000 then one 100000 batch reaches max_samples = 25000


def test_mc_scenes_vs_oracle_early_stop(eng, oracle, wl, pkg):
    """Scenes placed so that most stop early (p ~ 0.5 needs ~1e4 samples): exercises the
    compaction of survivors and the per-scene stop decision."""
    poses, sds, _ = wl.random_tables(40, 40, seed=31)
    rng = np.random.default_rng(2)
    n = 300
    scenes = np.empty(n, pkg.SCENE_DT)
    scenes["x"] = rng.uniform(-4, 4, n)
    scenes["y"] = rng.uniform(-3, 3, n)
    scenes["var_idx"] = rng.integers(0, 40, n)
    scenes["pose_idx"] = rng.integers(0, 40, n)
    ref_h, ref_u, ref_rows, ref_total = oracle.mc_scenes(poses, sds, scenes, W, H, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 15000, 5, 700)
    hits, used, rows, total, _ = run_scenes(eng, pkg, poses, sds, scenes, 15000, 5, base=700)
    assert len(np.unique(ref_u)) > 3, "test no longer exercises several stop times"
    assert np.array_equal(used, ref_u) and np.array_equal(hits, ref_h)
    assert np.array_equal(rows.view(np.uint32), ref_rows.view(np.uint32))
    assert total == ref_total


def test_mc_scenes_sharding_is_invisible(eng, wl, pkg):
    """Evaluating scenes [0,n) in one call equals evaluating two shards with the matching
    scene_id_base — the multi-GPU partition of SURVEY.md §8e."""
    poses, sds, _ = wl.random_tables(64, 64, seed=41)
    d_p, d_s = eng.to_device(poses), eng.to_device(sds)
    n = 5000
    d_sc = eng.empty(n, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, W, H, 4.0, 3, 0, n, d_sc)
    scenes = d_sc.get()
    h, u, r, tot, _ = run_scenes(eng, pkg, poses, sds, scenes, 3000, 8)
    cut = 1777
    h0, u0, r0, t0, _ = run_scenes(eng, pkg, poses, sds, scenes[:cut], 3000, 8, base=0)
    h1, u1, r1, t1, _ = run_scenes(eng, pkg, poses, sds, scenes[cut:], 3000, 8, base=cut)
    assert np.array_equal(h, np.concatenate([h0, h1])) and np.array_equal(u, np.concatenate([u0, u1]))
    assert tot == t0 + t1


def test_mc_differential_fuzz(eng):
    """30 random configurations of tables, scene counts, robot sizes, accuracy bins, max_samples and sampling schedules:
    sampled scenes, per-scene hit / sample counts, output rows and one sample-parallel range per configuration equal the
    oracle's bit for bit (tests/tools/mc_fuzz.py runs the same generator for as many configurations as wanted)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "mc_fuzz.py")
    spec = importlib.util.spec_from_file_location("mc_fuzz", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(99)
    schedules = set()
    for i in range(30):
        ok, info = fz.one(eng, rng, i)
        assert ok, info
        schedules.add(info[1])
    assert len(schedules) >= 4


NAN, INF = float("nan"), float("inf")


@pytest.mark.parametrize("robot,pos,pose,sd", [
    ((4.07, 1.74), (NAN, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),    # NaN robot position: every robot vertex NaN
    ((4.07, 1.74), (3.0, INF), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),    # infinite robot position
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, NAN), (0.3, 0.3, 0.2, 0.0, 0.0)),    # NaN robot angle
    ((4.07, 1.74), (3.0, 1.0), (INF, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),    # infinite obstacle width
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (INF, 0.3, 0.2, 0.0, 0.0)),    # infinite sigma_x
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, NAN, 0.0, 0.0)),    # NaN sigma_theta
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, INF)),    # infinite sigma_h (third Box-Muller pair in use)
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (3e30, 3e30, 0.2, 0.0, 0.0)),  # finite but products overflow
    ((1e20, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),    # huge robot
    ((4.07, 1.74), (9e14, 1.0), (2.0, 1.0, 0.6), (0.3, 0.3, 0.2, 0.0, 0.0)),   # just inside the tame domain: the fast paths
    ((4.07, 1.74), (3.0, 1.0), (2.0, 1.0, 0.6), (9e14, 0.3, 0.2, 0.0, 0.0)),   # just inside, huge noise
])
def test_mc_pair_non_finite_scenes(eng, oracle, robot, pos, pose, sd):
    """Scene parameters outside the tame domain (include/c2d.h "non-finite inputs"): no pretest, the axis test that is
    defined for every bit pattern; hit counts equal the oracle's, for unaligned sample ranges too."""
    for begin, n in ((0, 50_000), (1_000_003, 4_097)):
        d = eng.zeros(1, np.uint64)
        eng.mc_pair(robot[0], robot[1], pos, pose, sd, 5, 9, begin, n, d)
        assert int(d.get()[0]) == oracle.mc_pair(robot[0], robot[1], pos, pose, sd, 5, 9, begin, n)
        d.free()


@pytest.mark.parametrize("exponent", [-10, -14, -16, -19, -21, -22, -23, -26, -30])
def test_mc_pair_whole_scene_at_tiny_scales(eng, oracle, wl, exponent):
    """The config-3 scene scaled down until products of two lengths are denormal: the shortcuts' margins are RELATIVE rounding
    bounds, so a scene with a nonzero length below 1e-15 takes the plain path (at 1e-22 and 1e-23 the fast paths differed from
    the oracle by up to 20 % of the hits)."""
    sc = wl.MC_PAIR_SCENE
    scale = 10.0 ** exponent
    w, h = sc["robot_w"] * scale, sc["robot_h"] * scale
    pos = (sc["pos"][0] * scale, sc["pos"][1] * scale)
    pose = (sc["pose"][0] * scale, sc["pose"][1] * scale, sc["pose"][2])
    for j, sd in enumerate(((0.3 * scale, 0.3 * scale, 0.2, 0.0, 0.0), (0.3 * scale, 0.3 * scale, 0.2, 0.05 * scale, 0.1 * scale), (3.0 * scale, 3.0 * scale, 0.2, 0.0, 0.0))):
        d = eng.zeros(1, np.uint64)
        eng.mc_pair(w, h, pos, pose, sd, 9, j, 1, 30_001, d)
        with np.errstate(all="ignore"):
            assert int(d.get()[0]) == oracle.mc_pair(w, h, pos, pose, sd, 9, j, 1, 30_001), (exponent, j)
        d.free()


def test_mc_scenes_with_non_finite_table_entries(eng, oracle, wl, pkg):
    """Adaptive loop over tables in which some poses / standard deviations are NaN, infinite or huge: those scenes take
    the plain path inside the same launches, the others are untouched; hits, sample counts and rows equal the oracle's."""
    poses, sds, _ = wl.random_tables(40, 40, seed=31, shape_variance=True)
    poses, sds = poses.copy(), sds.copy()
    poses["width"][3], poses["theta"][7], poses["height"][11] = np.nan, np.inf, 1e30
    sds["x"][2], sds["theta"][5], sds["height"][9], sds["y"][13] = np.inf, np.nan, np.inf, 1e20
    rng = np.random.default_rng(2)
    n = 400
    scenes = np.empty(n, pkg.SCENE_DT)
    scenes["x"] = rng.uniform(-4, 4, n)
    scenes["y"] = rng.uniform(-3, 3, n)
    scenes["var_idx"] = rng.integers(0, 40, n)
    scenes["pose_idx"] = rng.integers(0, 40, n)
    scenes["x"][17], scenes["y"][33] = np.nan, -np.inf
    ref_h, ref_u, ref_rows, ref_total = oracle.mc_scenes(poses, sds, scenes, W, H, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 25000, 5, 700)
    hits, used, rows, total, _ = run_scenes(eng, pkg, poses, sds, scenes, 25000, 5, base=700)
    assert np.array_equal(used, ref_u) and np.array_equal(hits, ref_h) and total == ref_total
    assert np.array_equal(rows.view(np.uint32), ref_rows.view(np.uint32))


@pytest.mark.parametrize("pos,pose,sd", [
    ((3.035, 0.0), (2.0, 1.0, 0.0), (1e-6, 1e-6, 0.0, 0.0, 0.0)),       # edges touch along x (|gap| ~ 1e-6): overlaps on robot axis 0 are a few ulps
    ((3.035, 0.0), (2.0, 1.0, 0.0), (1e-7, 0.0, 1e-7, 0.0, 0.0)),       # the same with a trembling angle
    ((0.5, 1.37), (2.0, 1.0, 0.0), (0.0, 1e-6, 0.0, 0.0, 0.0)),         # touching along y
    ((0.5, 1.37), (2.0, 1.0, 0.0), (0.3, 3e-7, 0.0, 0.0, 0.0)),
    ((3.0541454553603, 0.5), (2.0, 1.0, 0.7853982), (2e-7, 2e-7, 0.0, 0.0, 0.0)),   # the same, an order of magnitude closer
    ((3.035, 0.0), (2.0, 1.0, 0.0), (1e-5, 1e-5, 1e-5, 1e-5, 1e-5)),    # shape noise too
    ((3.0, 1.0), (2.0, 1.0, 0.3), (300.0, 300.0, 0.3, 0.0, 0.0)),        # large coordinates: the certificates ask for wide overlaps
    ((3.0541454553603, 0.5), (2.0, 1.0, 0.7853982), (1e-6, 1e-6, 1e-6, 0.0, 0.0)),   # a corner of the robot (at 45 degrees) grazing the obstacle
    ((3.035, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 0.5, 0.0, 0.0)),         # point obstacle: every obstacle axis is (0, 0)
    ((1e-20, 1e-20), (1e-19, 1e-19, 0.3), (1e-20, 1e-20, 0.2, 0.0, 0.0)),  # everything tiny: products underflow
])
def test_mc_parallel_axis_certificates_on_razor_thin_overlaps(eng, oracle, pos, pose, sd):
    """sample_collides_mask leaves out the second axis of each parallel pair when the overlap on the first is provably wide
    enough (c2d_mc.hip).  Scenes whose overlaps sit within a few ulps of zero force the full evaluation of those axes; the hit
    counts stay the oracle's, for the robot at the origin (exactly opposite edge axes) and rotated (rounded ones)."""
    for robot_theta_pos in (pos, (pos[0] * 0.8 - pos[1] * 0.6, pos[0] * 0.6 + pos[1] * 0.8)):
        for begin, n in ((0, 40_000), (777, 3_333)):
            d = eng.zeros(1, np.uint64)
            eng.mc_pair(W, H, robot_theta_pos, pose, sd, 21, 3, begin, n, d)
            assert int(d.get()[0]) == oracle.mc_pair(W, H, robot_theta_pos, pose, sd, 21, 3, begin, n), (robot_theta_pos, begin)
            d.free()


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1.0, 40.0), (1e-3, 0.0), (1e3, 0.0)])
def test_mc_closed_form_on_touching_scenes(eng, oracle, wl, scale, offset):
    """The full evaluation decides a sample by the sign of the closed-form gap when it exceeds a proven margin and by the
    reference's vertex arithmetic otherwise (c2d_mc.hip model_gap).  Scenes whose robot and obstacle TOUCH along one of the four
    frame directions (workloads.touching_pose_pairs, moved into the obstacle's frame), drawn with standard deviations of
    1e-7 .. 1e-4 of the scene's size, keep nearly every sample at the margin: the hit counts must stay the oracle's."""
    pairs = wl.touching_pose_pairs(48, seed=int(offset) + 5, scale=scale, offset=offset).astype(np.float64)
    rng = np.random.default_rng(3)
    for i in range(pairs.shape[1]):
        x1, y1, w1, h1, t1, x2, y2, w2, h2, t2 = pairs[:, i]
        c, s = np.cos(-t2), np.sin(-t2)
        pos = (float(c * (x1 - x2) - s * (y1 - y2)), float(s * (x1 - x2) + c * (y1 - y2)))
        sg = float(rng.choice([1e-7, 1e-6, 1e-5, 1e-4])) * scale
        sd = (sg, sg * float(rng.choice([0.0, 1.0])), float(rng.choice([0.0, 1e-7, 1e-5])), sg * float(rng.choice([0.0, 0.0, 1.0])), 0.0)
        pose = (float(w2), float(h2), float(t1 - t2))
        d = eng.zeros(1, np.uint64)
        eng.mc_pair(float(w1), float(h1), pos, pose, sd, 31, i, 0, 20_000, d)
        assert int(d.get()[0]) == oracle.mc_pair(float(w1), float(h1), pos, pose, sd, 31, i, 0, 20_000), (i, pos, pose, sd)
        d.free()


def test_canonical_math_over_whole_domains(eng, oracle):
    """Device == oracle, bit for bit, on EVERY input the Box-Muller transform can see: all 2^32 angle words, every float in
    [2^-33, 1] for the logarithm and in [2^-24, 64] for the square root (tests/tools/math_exhaustive.py holds the full set,
    float angles and the transform itself included: profiles/r03_math_exhaustive.txt)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("math_exhaustive", os.path.join(os.path.dirname(__file__), "tools", "math_exhaustive.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    assert tool.sweep(eng, eng.MATH_SINCOS_U32, "sin/cos of 2 pi y / 2^32, every 32-bit word y", [(0, 0xFFFFFFFF)]) == 0
    assert tool.sweep(eng, eng.MATH_LOG, "log(u), every float in [2^-33, 1]", [(tool.fbits(2.0 ** -33), tool.fbits(1.0))]) == 0
    assert tool.sweep(eng, eng.MATH_SQRT, "sqrt(v), every float in [2^-24, 64]", [(tool.fbits(2.0 ** -24), tool.fbits(64.0))]) == 0
