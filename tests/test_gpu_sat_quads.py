"""General quadrilaterals on the vertex-format kernels (utils.cu:159-184: convex_collide takes ANY two float[8], the kernels'
entry points any 16 floats per pair).  The kernels evaluate four of the eight edge axes and certify the other four from the
overlaps they saw (`rect_collide_certified`, csrc/c2d_math.hpp): for a rectangle edges 2, 3 are the negatives of edges 0, 1 up
to rounding, for a general quadrilateral they are not, and the certificate's `d = a + b` term is what keeps it sound.  The
proof covers every input; until this file the tests fed it rectangles and random exponents only.  Unit scale, three families:
random convex quadrilaterals, near-parallelograms whose fourth vertex is off by 0, ±1, ±4, ±64 ulp and whose partner nearly
touches them, and exactly touching configurations on a binary grid (every tie of the strict `<` of utils.cu:178 is exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ULPS = (0, 1, -1, 4, -4, 64, -64)


def _nudge(x, k):
    """x moved by k units in the last place (float32)"""
    x = np.asarray(x, np.float32)
    return (x + np.float32(k) * np.spacing(np.abs(x)).astype(np.float32)).astype(np.float32)


def random_convex_quads(n, rng):
    """[8][n]: four points on a random ellipse in angular order (half of them clockwise), rotated and moved"""
    ang = np.sort(rng.uniform(0, 2 * np.pi, (4, n)), axis=0)
    a, b = rng.uniform(0.3, 2.5, n), rng.uniform(0.3, 2.5, n)
    rot = rng.uniform(0, 2 * np.pi, n)
    cx, cy = rng.uniform(-3, 3, n), rng.uniform(-3, 3, n)
    x, y = a * np.cos(ang), b * np.sin(ang)
    X = np.cos(rot) * x - np.sin(rot) * y + cx
    Y = np.sin(rot) * x + np.cos(rot) * y + cy
    cw = rng.random(n) < 0.5
    X[:, cw], Y[:, cw] = X[::-1][:, cw], Y[::-1][:, cw]
    q = np.empty((8, n), np.float32)
    q[0::2], q[1::2] = X, Y
    return q


def near_parallelogram_pairs(n, rng):
    """A = p, p + u, p + u + v, p + v with the fourth vertex nudged by ULPS.  convex_collide's axes are the EDGE VECTORS
    (utils.cu:170-171), so what decides a near miss is a tie of the two projection intervals on an edge vector: B is a second
    near-parallelogram whose extreme vertex along one of A's edge vectors projects (to a few ulp, either side) onto the end of
    A's interval on that axis, the rest of B lying beyond it."""
    f = np.float32
    p = rng.uniform(-2, 2, (2, n)).astype(f)
    th = rng.uniform(0, 2 * np.pi, n)
    lu, lv = rng.uniform(0.5, 2.0, n), rng.uniform(0.5, 2.0, n)
    skew = rng.uniform(0.5, 2.6, n)                     # angle between u and v: parallelograms, not only rectangles
    u = np.stack([lu * np.cos(th), lu * np.sin(th)]).astype(f)
    v = np.stack([lv * np.cos(th + skew), lv * np.sin(th + skew)]).astype(f)
    A = np.empty((8, n), f)
    A[0:2] = p
    A[2:4] = p + u
    A[4:6] = (p + u) + v
    A[6:8] = p + v
    A[6], A[7] = _nudge(A[6], rng.choice(ULPS, n)), _nudge(A[7], rng.choice(ULPS, n))
    # the axis: A's edge 0 -> 1 or 1 -> 2 as the kernel forms it (a float difference of vertices); the end of A's interval on it
    e = rng.integers(0, 2, n)
    axis = np.where(e == 0, A[2:4] - A[0:2], A[4:6] - A[2:4]).astype(np.float64)
    side = np.where(rng.random(n) < 0.5, 1.0, -1.0)
    proj = np.stack([A[2 * k] * axis[0] + A[2 * k + 1] * axis[1] for k in range(4)])
    far = np.argmax(side * proj, axis=0)
    cols = np.arange(n)
    m = np.stack([A[2 * far, cols], A[2 * far + 1, cols]]).astype(np.float64)
    unit = axis / np.linalg.norm(axis, axis=0)
    perp = np.stack([-unit[1], unit[0]])
    q = (m + perp * rng.uniform(-1.2, 1.2, n)).astype(f)                     # same projection on the axis, up to rounding ...
    near_k = (0, 0, 1, -1, 2, -2, 8, -8, 300, -300)
    q[0], q[1] = _nudge(q[0], rng.choice(near_k, n)), _nudge(q[1], rng.choice(near_k, n))   # ... and a few ulp to either side
    # B = q, q + u2, q + u2 + v2, q + v2 with both edges pointing away from A along the axis
    d0 = np.arctan2(side * unit[1], side * unit[0])
    t1, t2 = d0 + rng.uniform(-1.3, 1.3, n), d0 + rng.uniform(-1.3, 1.3, n)
    l2u, l2v = rng.uniform(0.3, 1.5, n), rng.uniform(0.3, 1.5, n)
    u2 = np.stack([l2u * np.cos(t1), l2u * np.sin(t1)]).astype(f)
    v2 = np.stack([l2v * np.cos(t2), l2v * np.sin(t2)]).astype(f)
    B = np.empty((8, n), f)
    B[0:2] = q
    B[2:4] = q + u2
    B[4:6] = (q + u2) + v2
    B[6:8] = q + v2
    B[6], B[7] = _nudge(B[6], rng.choice(ULPS, n)), _nudge(B[7], rng.choice(ULPS, n))
    roll = rng.integers(0, 4, n)                        # B's vertices start anywhere
    for r in range(1, 4):
        mk = roll == r
        B[:, mk] = np.roll(B[:, mk], 2 * r, axis=0)
    swap = rng.random(n) < 0.5                          # either order of the two arguments
    P = np.concatenate([A, B])
    P[:, swap] = np.concatenate([B, A])[:, swap]
    return P


def touching_grid_pairs(n, rng):
    """convex quadrilaterals (trapezoids, kites, parallelograms, rectangles) with coordinates on a grid of 1/8: every product
    and sum of the test is exact, so shared vertices, shared edges and a vertex on an edge are EXACT ties of the strict `<`."""
    f = np.float32
    shapes = np.array([
        [0, 0, 4, 0, 3, 2, 1, 2],      # trapezoid
        [0, 0, 2, -1, 4, 0, 2, 3],     # kite
        [0, 0, 3, 0, 4, 2, 1, 2],      # parallelogram
        [0, 0, 3, 0, 3, 2, 0, 2],      # rectangle
        [0, 0, 5, 1, 4, 3, 1, 2],      # irregular convex
        [0, 0, 1, 2, 4, 3, 5, 1],      # the same clockwise-ish (reordered, still convex)
    ], np.float64) / 2.0
    ia, ib = rng.integers(0, len(shapes), n), rng.integers(0, len(shapes), n)
    A = shapes[ia].T.copy()
    B = shapes[ib].T.copy()
    for Q in (A, B):                                    # exact quarter turns and mirror images keep everything on the grid
        turn = rng.integers(0, 4, n)
        for _ in range(3):
            m = turn > 0
            Q[0::2, m], Q[1::2, m] = -Q[1::2, m].copy(), Q[0::2, m].copy()
            turn = turn - 1
        mir = rng.random(n) < 0.5
        Q[0::2, mir] = -Q[0::2, mir]
    A[0::2] += rng.integers(-8, 9, n) / 8.0
    A[1::2] += rng.integers(-8, 9, n) / 8.0
    # move B so that its vertex j coincides with A's vertex i, or with the midpoint of A's edge i, plus a grid offset in {-1/8, 0, 1/8}^2
    i, j = rng.integers(0, 4, n), rng.integers(0, 4, n)
    cols = np.arange(n)
    ax, ay = A[2 * i, cols], A[2 * i + 1, cols]
    nx_, ny_ = A[2 * ((i + 1) % 4), cols], A[2 * ((i + 1) % 4) + 1, cols]
    mid = rng.random(n) < 0.5
    tx, ty = np.where(mid, (ax + nx_) / 2, ax), np.where(mid, (ay + ny_) / 2, ay)
    dx = tx - B[2 * j, cols] + rng.integers(-1, 2, n) / 8.0
    dy = ty - B[2 * j + 1, cols] + rng.integers(-1, 2, n) / 8.0
    B[0::2] += dx
    B[1::2] += dy
    P = np.concatenate([A, B]).astype(f)
    assert np.array_equal(P.astype(np.float64), np.concatenate([A, B])), "the grid is not exact in float32"
    return P


def _all_entry_points(eng, planes, ref, ref_cnt, what):
    n = planes.shape[1]
    d = eng.to_device(planes)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts([d.row(k) for k in range(16)], n, d_out, d_cnt)
    got = d_out.get()
    assert np.array_equal(got, ref), "%s, vertex planes: %d of %d booleans differ, first at %d" % (what, (got != ref).sum(), n, int(np.argmax(got != ref)))
    assert int(d_cnt.get()[0]) == ref_cnt
    # unaligned planes: the one-pair-per-lane instance (all eight axes, no certificate)
    host = np.zeros((16, n + 4), np.float32)
    host[:, 1:n + 1] = planes
    d_un = eng.to_device(host)
    d_out2 = eng.zeros(n + 4, np.uint8)
    eng.sat_rect_pairs_verts([d_un.row(k) + 4 for k in range(16)], n, d_out2.ptr + 1, None)
    assert np.array_equal(d_out2.get()[1:n + 1], ref), what + ", unaligned planes"
    d_mask, d_cnt2 = eng.zeros((n + 63) // 64, np.uint64), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts_mask([d.row(k) for k in range(16)], n, d_mask, d_cnt2)
    bits = np.unpackbits(d_mask.get().view(np.uint8), bitorder="little")[:n]
    assert np.array_equal(bits, ref) and int(d_cnt2.get()[0]) == ref_cnt, what + ", bit-mask output"
    d1, d2 = eng.to_device(np.ascontiguousarray(planes[:8].T)), eng.to_device(np.ascontiguousarray(planes[8:].T))
    d_out3, d_cnt3 = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_aos(d1, d2, n, d_out3, d_cnt3)
    assert np.array_equal(d_out3.get(), ref) and int(d_cnt3.get()[0]) == ref_cnt, what + ", array-of-rectangles layout"
    for a in (d, d_out, d_cnt, d_un, d_out2, d_mask, d_cnt2, d1, d2, d_out3, d_cnt3):
        a.free()


def test_random_convex_quadrilaterals_at_unit_scale(eng, oracle):
    n = 1_000_000
    rng = np.random.default_rng(0xC0DE)
    planes = np.concatenate([random_convex_quads(n, rng), random_convex_quads(n, rng)])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    assert 0.15 < ref.mean() < 0.85, ref.mean()
    _all_entry_points(eng, planes, ref, ref_cnt, "random convex quadrilaterals")
    # spot check of the oracle's batch entry point against its per-pair convex_collide (utils.cu:159-184) on the first pairs
    for i in range(50):
        assert oracle.convex_collide(planes[:8, i], planes[8:, i]) == ref[i]


def test_near_parallelograms_with_a_nudged_fourth_vertex(eng, oracle):
    n = 400_000
    planes = near_parallelogram_pairs(n, np.random.default_rng(0xFACE))
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    assert 0.05 < ref.mean() < 0.8, ref.mean()
    moved = planes.copy()
    moved[8:] = _nudge(moved[8:], 4)
    flipped = (oracle.sat_rect_pairs_verts(moved)[0] != ref).mean()
    assert flipped > 0.02, "the set no longer sits on the razor's edge: %.4f of the booleans move under a 4-ulp nudge" % flipped
    _all_entry_points(eng, planes, ref, ref_cnt, "near-parallelograms")
    _all_entry_points(eng, moved, *oracle.sat_rect_pairs_verts(moved), "near-parallelograms, second quadrilateral moved by 4 ulp")


def test_touching_quadrilaterals_on_a_binary_grid(eng, oracle):
    n = 200_000
    planes = touching_grid_pairs(n, np.random.default_rng(0xBEAD))
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    assert 0.2 < ref.mean() < 0.98, ref.mean()
    _all_entry_points(eng, planes, ref, ref_cnt, "touching quadrilaterals")
    # exact touching counts as colliding (strict `<`, utils.cu:178): a pair sharing exactly one vertex, from the construction
    a = np.array([0, 0, 2, 0, 1.5, 1, 0.5, 1], np.float32)          # trapezoid
    b = np.array([2, 0, 4, -1, 4, 1, 3, 2], np.float32)             # touches it at (2, 0) only
    assert oracle.convex_collide(a, b) == 1
    one = np.concatenate([a, b])[:, None].repeat(4, axis=1)
    ref1, _ = oracle.sat_rect_pairs_verts(one)
    assert ref1.all()
    _all_entry_points(eng, one, ref1, 4, "one shared vertex")
