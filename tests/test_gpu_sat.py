"""GPU parity tests of the SAT kernels, through the C-ABI (libc2d.so), against
the CPU oracle on identical inputs.  Booleans and vertex bits must be equal."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_verts(eng, planes, with_count=True, offset_elems=0):
    """planes: float32 [16][n] host.  offset_elems shifts every plane and the
    output by that many elements to exercise the unaligned path."""
    n = planes.shape[1]
    d = eng.empty((16, n + 8), np.float32)
    host = np.zeros((16, n + 8), np.float32)
    host[:, offset_elems:offset_elems + n] = planes
    d2 = eng.to_device(host)
    d_out = eng.zeros(n + 16, np.uint8)
    d_cnt = eng.zeros(1, np.uint64)
    ptrs = [d2.row(k) + 4 * offset_elems for k in range(16)]
    eng.sat_rect_pairs_verts(ptrs, n, d_out.ptr + offset_elems, d_cnt if with_count else None)
    out = d_out.get()
    cnt = int(d_cnt.get()[0])
    for a in (d, d2, d_out, d_cnt):
        a.free()
    assert not out[:offset_elems].any() and not out[offset_elems + n:].any(), "wrote outside [0, n)"
    return out[offset_elems:offset_elems + n], cnt


def test_golden_rect_1k(eng):
    g = np.load(os.path.join(GOLD, "sat_rect_1k.npz"))
    out, cnt = run_verts(eng, g["planes"])
    assert np.array_equal(out, g["expected"])
    assert cnt == int(g["expected"].sum())


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 1023, 1025, 4099])
def test_ragged_sizes(eng, oracle, wl, n):
    poses = wl.random_obb_pose_planes(n, seed=100 + n, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    out, cnt = run_verts(eng, planes)
    assert np.array_equal(out, ref) and cnt == ref_cnt


@pytest.mark.parametrize("offset", [1, 2, 3])
def test_unaligned_buffers(eng, oracle, wl, offset):
    n = 2049
    poses = wl.random_obb_pose_planes(n, seed=9, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    out, cnt = run_verts(eng, planes, offset_elems=offset)
    assert np.array_equal(out, ref) and cnt == ref_cnt


def test_empty_and_null(eng, pkg):
    eng.sat_rect_pairs_verts([0] * 16, 0, 0, None)  # n == 0 is a no-op
    with pytest.raises(pkg.C2DError):
        eng.sat_rect_pairs_verts([0] * 16, 10, 0, None)


def test_count_accumulates_and_is_optional(eng, oracle, wl):
    n = 10000
    poses = wl.random_obb_pose_planes(n, seed=5, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    d = eng.to_device(planes)
    d_out = eng.empty(n, np.uint8)
    d_cnt = eng.zeros(1, np.uint64)
    ptrs = [d.row(k) for k in range(16)]
    eng.sat_rect_pairs_verts(ptrs, n, d_out, None)
    assert np.array_equal(d_out.get(), ref)
    for _ in range(3):
        eng.sat_rect_pairs_verts(ptrs, n, d_out, d_cnt)
    assert int(d_cnt.get()[0]) == 3 * ref_cnt


def test_random_1m_verts_and_pose_paths(eng, oracle, wl):
    """1e6 seeded pairs (the config-2 distribution): rects_from_poses bit-exact,
    then vertex-format and pose-format SAT bit-exact."""
    n = 1_000_003
    poses = wl.random_obb_pose_planes(n, seed=0x5A7)
    d_pose = eng.to_device(poses)
    d_planes = eng.empty((16, n), np.float32)
    for r in range(2):
        eng.rects_from_poses(*[d_pose.row(5 * r + k) for k in range(5)], n, [d_planes.row(8 * r + k) for k in range(8)])
    planes = d_planes.get()
    ref_planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    assert np.array_equal(planes.view(np.uint32), ref_planes.view(np.uint32))
    ref, ref_cnt = oracle.sat_rect_pairs_verts(ref_planes)
    d_out = eng.empty(n, np.uint8)
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts([d_planes.row(k) for k in range(16)], n, d_out, d_cnt)
    assert np.array_equal(d_out.get(), ref) and int(d_cnt.get()[0]) == ref_cnt
    d_out2 = eng.zeros(n, np.uint8)
    d_cnt2 = eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_pose([d_pose.row(k) for k in range(10)], n, d_out2, d_cnt2)
    assert np.array_equal(d_out2.get(), ref) and int(d_cnt2.get()[0]) == ref_cnt
    ref_pose, _ = oracle.sat_rect_pairs_pose(poses)
    assert np.array_equal(ref_pose, ref)


@pytest.mark.parametrize("n", [1, 255, 100_003])
def test_aos_layout_matches_planes(eng, oracle, wl, n):
    """convex_collide's own layout: float[8] per rectangle, f32[n][8] arrays."""
    poses = wl.random_obb_pose_planes(n, seed=300 + n, extent=3.0)
    r1, r2 = oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(np.concatenate([r1, r2]))
    d1, d2 = eng.to_device(np.ascontiguousarray(r1.T)), eng.to_device(np.ascontiguousarray(r2.T))
    d_out, d_cnt = eng.zeros(n + 8, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_aos(d1, d2, n, d_out, d_cnt)
    out = d_out.get()
    assert np.array_equal(out[:n], ref) and not out[n:].any() and int(d_cnt.get()[0]) == ref_cnt
    for i in range(min(n, 5)):
        assert oracle.convex_collide(r1[:, i], r2[:, i]) == ref[i]


def test_razor_edge_pairs(eng, oracle):
    """Pairs built to sit on the decision boundary: touching edges shifted by
    0, +-1, +-2 ulp at several scales and angles.  This is where a fused
    multiply-add or a different min/max would flip booleans."""
    rng = np.random.default_rng(12)
    n = 20000
    w1, h1, w2, h2 = (rng.uniform(0.5, 3, n).astype(np.float32) for _ in range(4))
    th = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    cx = rng.uniform(-50, 50, n).astype(np.float32)
    cy = rng.uniform(-50, 50, n).astype(np.float32)
    # second rectangle shares the orientation and touches along the first one's +x edge
    # separation = touching distance +- k coordinate-ulps (ulp(64) = 2^-17 covers |c| < 64)
    kk = (rng.integers(-8, 9, n) * 0.25).astype(np.float32)
    gap = ((w1 + w2) / 2 + kk * np.float32(2.0**-17)).astype(np.float32)
    cx2 = (cx + gap * np.cos(th)).astype(np.float32)
    cy2 = (cy + gap * np.sin(th)).astype(np.float32)
    planes = np.concatenate([oracle.rects_from_poses(cx, cy, w1, h1, th), oracle.rects_from_poses(cx2, cy2, w2, h2, th)])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    assert 0.2 < ref.mean() < 0.8, "generator no longer straddles the boundary"
    out, cnt = run_verts(eng, planes)
    assert np.array_equal(out, ref) and cnt == ref_cnt


def test_extreme_magnitudes_and_denormals(eng, oracle):
    """Vertices spanning 2^-149 .. 2^60 (subnormals included), plus exact zeros and signed zeros:
    the GPU must keep subnormals (no flush-to-zero) and order +-0 like the oracle does."""
    rng = np.random.default_rng(99)
    n = 200_000
    mag = np.exp2(rng.uniform(-149, 60, (16, n)))
    planes = (mag * rng.choice([-1.0, 1.0], (16, n))).astype(np.float32)
    planes[:, : n // 10][rng.random((16, n // 10)) < 0.3] = 0.0
    planes[:, n // 10: n // 5][rng.random((16, n // 5 - n // 10)) < 0.3] = -0.0
    # a block where everything is subnormal or tiny, so that products underflow
    planes[:, n // 2: n // 2 + 20000] = (np.exp2(rng.uniform(-149, -120, (16, 20000))) * rng.choice([-1.0, 1.0], (16, 20000))).astype(np.float32)
    assert np.isfinite(planes).all() and (np.abs(planes[planes != 0]) < 1e-38).any()
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    out, cnt = run_verts(eng, planes)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    assert 0 < ref_cnt < n


def test_full_size_properties_1e7(eng, wl):
    """BASELINE config 2 size (1e7 pairs): size-independent properties.
    collide(a,b) == collide(b,a); collide is invariant under a cyclic shift of
    either vertex list; the device count equals the sum of the booleans."""
    n = 10_000_000
    poses = wl.random_obb_pose_planes(n, seed=0x5A7)
    d_pose = eng.to_device(poses)
    del poses
    d_planes = eng.empty((16, n), np.float32)
    for r in range(2):
        eng.rects_from_poses(*[d_pose.row(5 * r + k) for k in range(5)], n, [d_planes.row(8 * r + k) for k in range(8)])
    d_pose.free()
    rows = [d_planes.row(k) for k in range(16)]
    d_ab, d_ba, d_sh = eng.empty(n, np.uint8), eng.empty(n, np.uint8), eng.empty(n, np.uint8)
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts(rows, n, d_ab, d_cnt)
    eng.sat_rect_pairs_verts(rows[8:] + rows[:8], n, d_ba, None)
    shifted = rows[2:8] + rows[0:2] + rows[8:]
    eng.sat_rect_pairs_verts(shifted, n, d_sh, None)
    ab, ba, sh = d_ab.get(), d_ba.get(), d_sh.get()
    assert set(np.unique(ab).tolist()) <= {0, 1}
    assert np.array_equal(ab, ba)
    assert np.array_equal(ab, sh)
    assert int(d_cnt.get()[0]) == int(ab.sum(dtype=np.int64))
    assert 0.05 < ab.mean() < 0.2
    for a in (d_planes, d_ab, d_ba, d_sh, d_cnt):
        a.free()


# ---- polygons ---------------------------------------------------------------------------------

def run_poly(eng, vx, vy, k):
    n = vx.shape[-1]
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    d_out = eng.zeros(n + 8, np.uint8)
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_poly_pairs(dvx, dvy, dk, n, d_out, d_cnt)
    out, cnt = d_out.get(), int(d_cnt.get()[0])
    for a in (dvx, dvy, dk, d_out, d_cnt):
        a.free()
    assert not out[n:].any()
    return out[:n], cnt


def test_poly_golden(eng):
    g = np.load(os.path.join(GOLD, "poly_k16_1k.npz"))
    out, cnt = run_poly(eng, g["vx"], g["vy"], g["k"])
    assert np.array_equal(out, g["expected"]) and cnt == int(g["expected"].sum())


@pytest.mark.parametrize("n,kmin,kmax", [(1, 3, 16), (7, 3, 16), (64, 16, 16), (65, 1, 2), (1000, 3, 5), (200001, 3, 16)])
def test_poly_random(eng, oracle, wl, n, kmin, kmax):
    vx, vy, k = wl.random_convex_polygons(n, seed=n, kmin=kmin, kmax=kmax, extent=3.0)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    out, cnt = run_poly(eng, vx, vy, k)
    assert np.array_equal(out, ref) and cnt == ref_cnt


def test_poly_bad_vertex_count_is_an_error(eng, pkg, wl):
    vx, vy, k = wl.random_convex_polygons(100, seed=1)
    k[1, 37] = 17
    with pytest.raises(pkg.C2DError):
        run_poly(eng, vx, vy, k)
    k[1, 37] = 0
    with pytest.raises(pkg.C2DError):
        run_poly(eng, vx, vy, k)
