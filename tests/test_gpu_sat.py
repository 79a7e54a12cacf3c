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


@pytest.mark.parametrize("n,out_offset", [(1, 0), (2, 0), (126, 0), (127, 0), (128, 0), (129, 0), (255, 0), (257, 1), (100_003, 0), (100_004, 1)])
def test_aos_layout_matches_planes(eng, oracle, wl, n, out_offset):
    """convex_collide's own layout: float[8] per rectangle, f32[n][8] arrays; 128-pair tiles with ragged ends; an
    odd-aligned output buffer takes the byte-store instance."""
    poses = wl.random_obb_pose_planes(n, seed=300 + n, extent=3.0)
    r1, r2 = oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(np.concatenate([r1, r2]))
    d1, d2 = eng.to_device(np.ascontiguousarray(r1.T)), eng.to_device(np.ascontiguousarray(r2.T))
    d_out, d_cnt = eng.zeros(n + 8, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_aos(d1, d2, n, d_out.ptr + out_offset, d_cnt)
    out = d_out.get()
    assert not out[:out_offset].any()
    out = out[out_offset:]
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


def test_full_size_1e7_oracle_equality_and_properties(eng, oracle, wl):
    """BASELINE config 2 size (1e7 pairs): every boolean equals the oracle's ("boolean equality on the full
    10^7 set"; the OpenMP oracle needs ~50 ms for it), plus the size-independent properties:
    collide(a,b) == collide(b,a); collide is invariant under a cyclic shift of
    either vertex list; the device count equals the sum of the booleans."""
    n = 10_000_000
    poses = wl.random_obb_pose_planes(n, seed=int(os.environ.get("C2D_FULLSIZE_SEED", "0x5A7"), 0))   # (another seed: profiles/r06_fullsize_seeds.sh)
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
    ref, ref_cnt = oracle.sat_rect_pairs_verts(d_planes.get())   # all 1e7 pairs on the CPU
    assert ref.shape == ab.shape == (n,)
    assert np.array_equal(ab, ref) and ref_cnt == int(d_cnt.get()[0])
    for a in (d_planes, d_ab, d_ba, d_sh, d_cnt):
        a.free()


@pytest.mark.parametrize("n,offset", [(1, 0), (3, 0), (4, 0), (5, 0), (1023, 0), (100_001, 0), (4096, 1), (777, 3)])
def test_rects_from_poses_ragged_and_unaligned(eng, oracle, wl, n, offset):
    """create_rect + rot_trans_rectangle over SoA: the 4-rectangles-per-lane path, its tail, and planes that are not
    16-byte aligned (one rectangle per lane) all give the oracle's vertices bit for bit."""
    poses = wl.random_obb_pose_planes(n, seed=7 + n)[:5]
    host = np.zeros((5, n + 8), np.float32)
    host[:, offset:offset + n] = poses
    d_in = eng.to_device(host)
    d_out = eng.zeros((8, n + 8), np.float32)
    eng.rects_from_poses(*[d_in.row(k) + 4 * offset for k in range(5)], n, [d_out.row(k) + 4 * offset for k in range(8)])
    got = d_out.get()
    ref = oracle.rects_from_poses(*poses)
    assert np.array_equal(got[:, offset:offset + n].view(np.uint32), ref.view(np.uint32))
    assert not got[:, :offset].any() and not got[:, offset + n:].any()        # nothing written outside
    d_in.free()
    d_out.free()


@pytest.mark.parametrize("n,offset", [(1, 0), (63, 0), (64, 0), (65, 0), (255, 0), (256, 0), (257, 0), (1000, 0), (100_003, 0),
                                      (1_000_003, 0), (5000, 1), (70_001, 3)])
def test_bit_mask_output(eng, oracle, wl, n, offset):
    """c2d_sat_rect_pairs_verts_mask: one bit per pair, little-endian within 64-bit words, unused high bits zero;
    offset != 0 shifts the planes off their 16-byte alignment (one pair per lane path)."""
    poses = wl.random_obb_pose_planes(n, seed=31 + n, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    host = np.zeros((16, n + 8), np.float32)
    host[:, offset:offset + n] = planes
    d = eng.to_device(host)
    words = (n + 63) // 64
    d_mask = eng.to_device(np.full(words + 2, 0xFFFFFFFFFFFFFFFF, np.uint64))   # poisoned: every word must be written
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts_mask([d.row(k) + 4 * offset for k in range(16)], n, d_mask, d_cnt)
    mask = d_mask.get()
    assert (mask[words:] == 0xFFFFFFFFFFFFFFFF).all()                            # nothing written past the mask
    bits = np.unpackbits(mask[:words].view(np.uint8), bitorder="little")
    assert np.array_equal(bits[:n], ref) and not bits[n:].any()
    assert int(d_cnt.get()[0]) == ref_cnt
    for a in (d, d_mask, d_cnt):
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


@pytest.mark.parametrize("n,kmin,kmax,extent", [(63, 3, 16, 1.0), (64, 3, 16, 1.0), (129, 1, 16, 0.5), (5000, 3, 16, 0.7),
                                                 (100_003, 3, 16, 1.5), (30_000, 3, 4, 1.0), (30_000, 12, 16, 1.0)])
def test_poly_dense_scenes(eng, oracle, wl, n, kmin, kmax, extent):
    """Small extents: most pairs collide or nearly touch, so the wave-cooperative full evaluation
    (phase 2 of sat_poly_kernel) decides most of them."""
    vx, vy, k = wl.random_convex_polygons(n, seed=1000 + n, kmin=kmin, kmax=kmax, extent=extent)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    out, cnt = run_poly(eng, vx, vy, k)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    assert 0.2 < ref.mean() < 1.0


@pytest.mark.parametrize("rows,kmin,kmax,n,extent", [(4, 3, 4, 100_000, 8.0), (4, 1, 4, 40_000, 0.6), (4, 3, 4, 4, 1.0), (4, 2, 4, 260, 0.8), (4, 3, 3, 1_000_000, 2.0),
                                                     (4, 3, 4, 100_003, 8.0), (4, 1, 4, 30_001, 0.6), (3, 3, 3, 5000, 1.0), (8, 3, 8, 100_003, 8.0),
                                                     (8, 5, 8, 30_001, 0.8), (6, 3, 6, 20_000, 1.5), (12, 3, 12, 50_001, 2.0), (9, 9, 9, 777, 1.0),
                                                     (16, 3, 16, 10_000, 1.0), (1, 1, 1, 200, 1.0)])
def test_poly_row_layouts(eng, pkg, oracle, wl, rows, kmin, kmax, n, extent):
    """c2d_sat_poly_pairs_rows: layouts with fewer than C2D_POLY_KMAX vertex rows per polygon run the kernel instances sized
    for 4 / 8 / 16 slots (rows <= 4: eight pairs per wave in the full evaluation, rows <= 8: four); sparse and dense scenes.
    rows == 4 with n a multiple of 4 takes the register-only kernel (4 pairs per lane, all 8 axes, no second phase)."""
    vx, vy, k = wl.random_convex_polygons(n, seed=rows * 1000 + n, kmin=kmin, kmax=kmax, extent=extent, rows=rows)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    d_out, d_cnt = eng.zeros(n + 8, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_rows(dvx, dvy, dk, n, rows, d_out, d_cnt)
    out, cnt = d_out.get(), int(d_cnt.get()[0])
    assert np.array_equal(out[:n], ref) and cnt == ref_cnt and not out[n:].any()
    # a count above `rows` (but within KMAX) is an error of THIS layout
    if rows < wl.KMAX:
        k2 = k.copy()
        k2[0, n // 2] = rows + 1
        dk2 = eng.to_device(k2)
        eng.sat_poly_pairs_rows(dvx, dvy, dk2, n, rows, d_out, None)
        with pytest.raises(pkg.C2DError):
            eng.synchronize()
        dk2.free()
    with pytest.raises(pkg.C2DError):
        eng.sat_poly_pairs_rows(dvx, dvy, dk, n, 17, d_out, None)
    for a in (dvx, dvy, dk, d_out, d_cnt):
        a.free()


def test_poly_padding_is_never_interpreted_and_orientation_is_free(eng, oracle, wl):
    """Slots at and beyond the vertex count may hold anything (NaN, inf, huge values); clockwise polygons
    (inward-pointing (-ey, ex)) must give the oracle's booleans as well."""
    n = 40_000
    vx, vy, k = wl.random_convex_polygons(n, seed=5, extent=2.0)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    junk = np.array([np.nan, np.inf, -np.inf, 3e38, -1e-40], np.float32)
    rng = np.random.default_rng(3)
    for p in range(2):
        mask = np.arange(wl.KMAX)[:, None] >= k[p][None, :]
        vx[p][mask] = rng.choice(junk, size=int(mask.sum()))
        vy[p][mask] = rng.choice(junk, size=int(mask.sum()))
    out, cnt = run_poly(eng, vx, vy, k)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    # reverse the vertex order of every second polygon A and every third polygon B
    cw_x, cw_y = vx.copy(), vy.copy()
    for p, step in ((0, 2), (1, 3)):
        for kk in range(1, wl.KMAX + 1):
            sel = np.flatnonzero((k[p] == kk) & (np.arange(n) % step == 0))
            cw_x[p][:kk, sel] = vx[p][:kk, sel][::-1]
            cw_y[p][:kk, sel] = vy[p][:kk, sel][::-1]
    ref2, ref2_cnt = oracle.sat_poly_pairs(np.nan_to_num(cw_x, nan=0, posinf=0, neginf=0), np.nan_to_num(cw_y, nan=0, posinf=0, neginf=0), k)
    out2, cnt2 = run_poly(eng, cw_x, cw_y, k)
    assert np.array_equal(out2, ref2) and cnt2 == ref2_cnt


def test_poly_bad_vertex_count_is_reported_at_the_next_sync(eng, pkg, wl):
    """Vertex counts are checked inside the kernel (the call stays asynchronous and capturable): the pair is
    written as 0 and the error comes back from the next c2d_stream_synchronize / c2d_ctx_check_async, once."""
    vx, vy, k = wl.random_convex_polygons(100, seed=1)
    for bad in (17, 0, 255):
        k2 = k.copy()
        k2[1, 37] = bad
        with pytest.raises(pkg.C2DError) as ei:
            run_poly(eng, vx, vy, k2)
        assert ei.value.status == -1 and "vertex count" in str(ei.value)
        eng.check_async()  # reported once, then clear
    # the other pairs of such a batch are still evaluated; the bad one reads 0
    k2 = k.copy()
    k2[0, 5] = 200
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k2)
    d_out = eng.zeros(100, np.uint8)
    eng.sat_poly_pairs(dvx, dvy, dk, 100, d_out, None)
    with pytest.raises(pkg.C2DError):
        eng.synchronize()
    got = d_out.get()
    from oracle import cpu as oracle

    ref, _ = oracle.sat_poly_pairs(vx, vy, k)
    ref[5] = 0
    assert np.array_equal(got, ref)
    for a in (dvx, dvy, dk, d_out):
        a.free()


def test_workspace_guard_refuses_a_second_stream(eng, pkg, wl):
    """The ctx owns one workspace: a counting call on stream B while a call on stream A is still running
    returns C2D_ERR_UNSUPPORTED instead of corrupting both (include/c2d.h conventions)."""
    tp, ts, _ = wl.random_tables(64, 64, seed=2)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    ns = 200_000
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, 4.07, 1.74, 4.0, 1, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    planes = eng.to_device(np.zeros((16, 256), np.float32))
    d_out, d_cnt = eng.zeros(256, np.uint8), eng.zeros(1, np.uint64)
    eng.synchronize()
    sa, sb = eng.stream_create(), eng.stream_create()
    eng.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 400_000, 3, 0, d_h, d_u,
                        stream=sa)  # tens of milliseconds of queued work
    with pytest.raises(pkg.C2DError) as ei:
        eng.sat_rect_pairs_verts([planes.row(i) for i in range(16)], 256, d_out, d_cnt, stream=sb)
    assert ei.value.status == -5
    eng.sat_rect_pairs_verts([planes.row(i) for i in range(16)], 256, d_out, None, stream=sb)  # no count: no workspace
    eng.synchronize(sa)
    eng.sat_rect_pairs_verts([planes.row(i) for i in range(16)], 256, d_out, d_cnt, stream=sb)  # stream A is done
    eng.synchronize(sb)
    assert int(d_cnt.get()[0]) == 256
    eng.stream_destroy(sa)
    eng.stream_destroy(sb)
    for a in (d_p, d_s, d_sc, d_h, d_u, planes, d_out, d_cnt):
        a.free()


def test_poly_differential_fuzz(eng):
    """60 random (batch size, row layout, vertex-count range, density) configurations with clockwise polygons and junk in the
    padded slots against the oracle (tests/tools/poly_fuzz.py runs the same generator for as many configurations as wanted)."""
    import importlib.util

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "poly_fuzz.py")
    spec = importlib.util.spec_from_file_location("poly_fuzz", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(77)
    seen_rows = set()
    for i in range(60):
        ok, info = fz.one(eng, rng, i)
        assert ok, info
        seen_rows.add(info[0])
    assert len(seen_rows) >= 10


# ---- non-finite inputs (include/c2d.h "non-finite inputs"; reference utils.cu:176-178) -------------------------

def test_non_finite_vertices(eng, oracle, wl):
    """NaN / inf / overflowing coordinates: every rectangle entry point returns the oracle's booleans, i.e. the
    comparison-based extremes of thrust::minmax_element (a NaN first projection keeps an axis from separating, a NaN
    at a later vertex is skipped).  Wide (4 pairs per lane), scalar (unaligned), bit-mask and array-of-rectangles paths."""
    n = 200_003
    poses = wl.random_obb_pose_planes(n, seed=1, extent=2.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    bad = wl.inject_non_finite(planes, seed=3)
    ref, ref_cnt = oracle.sat_rect_pairs_verts(bad)
    fin, _ = oracle.sat_rect_pairs_verts(planes)
    assert (ref != fin).mean() > 0.05, "the injected values no longer change results"
    for off in (0, 1):
        out, cnt = run_verts(eng, bad, offset_elems=off)
        assert np.array_equal(out, ref) and cnt == ref_cnt
    d = eng.to_device(bad)
    words = (n + 63) // 64
    d_mask, d_cnt = eng.zeros(words, np.uint64), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts_mask([d.row(k) for k in range(16)], n, d_mask, d_cnt)
    bits = np.unpackbits(d_mask.get().view(np.uint8), bitorder="little")
    assert np.array_equal(bits[:n], ref) and int(d_cnt.get()[0]) == ref_cnt
    d1, d2 = eng.to_device(np.ascontiguousarray(bad[:8].T)), eng.to_device(np.ascontiguousarray(bad[8:].T))
    d_out = eng.zeros(n, np.uint8)
    eng.sat_rect_pairs_aos(d1, d2, n, d_out, None)
    assert np.array_equal(d_out.get(), ref)
    for a in (d, d_mask, d_cnt, d1, d2, d_out):
        a.free()


def test_non_finite_poses(eng, oracle, wl):
    """Pose format with NaN / inf / huge pose components: rectangles are rebuilt with the same IEEE operations on both
    sides, so vertices that are numbers agree bit for bit, NaNs sit in the same places and the booleans are equal."""
    n = 100_001
    poses = wl.inject_non_finite(wl.random_obb_pose_planes(n, seed=2, extent=2.0), seed=6, frac=0.3)
    ref, ref_cnt = oracle.sat_rect_pairs_pose(poses)
    d_pose = eng.to_device(poses)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_pose([d_pose.row(k) for k in range(10)], n, d_out, d_cnt)
    assert np.array_equal(d_out.get(), ref) and int(d_cnt.get()[0]) == ref_cnt
    d_planes = eng.empty((8, n), np.float32)
    eng.rects_from_poses(*[d_pose.row(k) for k in range(5)], n, [d_planes.row(k) for k in range(8)])
    got, want = d_planes.get(), oracle.rects_from_poses(*poses[:5])
    assert np.array_equal(np.isnan(got), np.isnan(want))
    num = ~np.isnan(want)
    assert np.array_equal(got[num].view(np.uint32), want[num].view(np.uint32))
    for a in (d_pose, d_out, d_cnt, d_planes):
        a.free()


@pytest.mark.parametrize("rows,kmin,kmax,n", [(16, 3, 16, 100_003), (12, 3, 12, 30_001), (8, 3, 8, 50_001), (4, 3, 4, 40_000), (4, 1, 4, 30_001)])
def test_non_finite_polygons(eng, oracle, wl, rows, kmin, kmax, n):
    """The same contract for the polygon kernels (every instance: 16 / 8 / 4 slots and the register-only 4-row kernel):
    non-finite coordinates in REAL vertices give the oracle's booleans; padded slots are never interpreted anyway."""
    vx, vy, k = wl.random_convex_polygons(n, seed=rows + n, kmin=kmin, kmax=kmax, extent=1.5, rows=rows)
    bx = wl.inject_non_finite(vx.reshape(2 * rows, -1), seed=4).reshape(vx.shape)
    by = wl.inject_non_finite(vy.reshape(2 * rows, -1), seed=5, frac=0.2).reshape(vy.shape)
    ref, ref_cnt = oracle.sat_poly_pairs(bx, by, k)
    fin, _ = oracle.sat_poly_pairs(vx, vy, k)
    assert (ref != fin).mean() > 0.01
    dvx, dvy, dk = eng.to_device(bx), eng.to_device(by), eng.to_device(k)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_rows(dvx, dvy, dk, n, rows, d_out, d_cnt)
    assert np.array_equal(d_out.get(), ref) and int(d_cnt.get()[0]) == ref_cnt
    for a in (dvx, dvy, dk, d_out, d_cnt):
        a.free()


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1.0, 60.0), (1e-4, 0.0), (1e3, 0.0), (1e-13, 0.0), (1e14, 0.0)])
def test_pose_closed_form_on_touching_pairs(eng, oracle, wl, scale, offset):
    """The pose-format kernel decides a pair by the closed-form gap when it exceeds a proven margin and by the vertex arithmetic
    otherwise (c2d_sat.hip pose_pair_closed_form).  Pairs built to touch along a frame direction, a few parts in 1e-7 .. 1e-3
    either side (workloads.touching_pose_pairs), at several scales — down to extents below the 1e-12 floor and up to the 1e15
    ceiling of the closed form — and far from the origin: every boolean is the oracle's."""
    n = 200_003
    poses = wl.touching_pose_pairs(n, seed=int(offset) + 9, scale=scale, offset=offset)
    ref, ref_cnt = oracle.sat_rect_pairs_pose(poses)
    assert 0.2 < ref.mean() < 0.8
    for off in (0, 1):  # 16-byte aligned planes (4 pairs per lane) and unaligned ones (the plain kernel)
        host = np.zeros((10, n + 4), np.float32)
        host[:, off:off + n] = poses
        d_pose = eng.to_device(host)
        rows = [d_pose.row(k) + 4 * off for k in range(10)]
        d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
        eng.sat_rect_pairs_pose(rows, n, d_out, d_cnt)
        assert np.array_equal(d_out.get(), ref) and int(d_cnt.get()[0]) == ref_cnt
        for a in (d_pose, d_out, d_cnt):
            a.free()
