"""GPU parity tests of the host-resident rectangle-pair entry points (c2d_sat_rect_pairs_verts_host / _pose_host: the pipelined
form of the reference's own upload - kernel - download, compute_collision_probability.cu:270-274, :314-318): booleans and counts
equal the CPU oracle's for pageable and page-locked buffers, ragged sizes, several chunks, unaligned plane starts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CHUNK = 1 << 24


def _planes(wl, oracle, n, seed):
    poses = wl.random_obb_pose_planes(max(n, 1), seed=seed)[:, :n]
    verts = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])]) if n else np.zeros((16, 0), np.float32)
    return np.ascontiguousarray(poses), np.ascontiguousarray(verts)


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 1 << 20, (1 << 22) + 3, CHUNK + 5])
def test_pageable_buffers_match_the_oracle(eng, oracle, wl, n):
    poses, verts = _planes(wl, oracle, n, seed=100 + n % 97)
    out = np.full(n + 8, 7, np.uint8)
    cnt = eng.sat_rect_pairs_host([verts[k] for k in range(16)], out[:n], "verts")
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts) if n else (np.zeros(0, np.uint8), 0)
    assert np.array_equal(out[:n], ref) and cnt == ref_cnt and (out[n:] == 7).all()
    out[:] = 7
    cnt = eng.sat_rect_pairs_host([poses[k] for k in range(10)], out[:n], "pose")
    ref, ref_cnt = oracle.sat_rect_pairs_pose(poses) if n else (np.zeros(0, np.uint8), 0)
    assert np.array_equal(out[:n], ref) and cnt == ref_cnt and (out[n:] == 7).all()


def test_page_locked_buffers_and_unaligned_planes(eng, oracle, wl):
    n = (1 << 21) + 4099
    poses, verts = _planes(wl, oracle, n, seed=5)
    # every plane starts one float further into its own page-locked array: 4-byte aligned sources, chunk boundaries anywhere
    hv = [eng.host_empty(n + 16, np.float32) for _ in range(16)]
    for k in range(16):
        hv[k][k:k + n] = verts[k]
    out = eng.host_empty(n, np.uint8)
    out[:] = 9
    cnt = eng.sat_rect_pairs_host([hv[k][k:k + n] for k in range(16)], out, "verts")
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    hp = [eng.host_empty(n, np.float32) for _ in range(10)]
    for k in range(10):
        hp[k][:] = poses[k]
    out[:] = 9
    cnt = eng.sat_rect_pairs_host(hp, out, "pose")
    ref, ref_cnt = oracle.sat_rect_pairs_pose(poses)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    # page-locked input, pageable output and the other way round
    out2 = np.zeros(n, np.uint8)
    assert eng.sat_rect_pairs_host(hp, out2, "pose") == ref_cnt and np.array_equal(out2, ref)
    assert eng.sat_rect_pairs_host([poses[k] for k in range(10)], out, "pose") == ref_cnt and np.array_equal(out, ref)
    for a in hv + hp + [out]:
        eng.host_free(a)


def test_argument_errors_and_device_entry_points_still_work_beside_it(eng, pkg, oracle, wl):
    import ctypes as C

    n = 5000
    poses, verts = _planes(wl, oracle, n, seed=9)
    out = np.zeros(n, np.uint8)
    arr = (C.c_void_p * 16)(*[verts[k].ctypes.data for k in range(15)] + [None])
    assert eng.lib.c2d_sat_rect_pairs_verts_host(eng.h, arr, n, C.c_void_p(out.ctypes.data), None) == -1     # a NULL plane
    arr = (C.c_void_p * 16)(*[verts[k].ctypes.data for k in range(16)])
    assert eng.lib.c2d_sat_rect_pairs_verts_host(eng.h, arr, n, None, None) == -1                             # NULL output
    assert eng.lib.c2d_sat_rect_pairs_verts_host(eng.h, arr, n, C.c_void_p(out.ctypes.data), None) == 0       # no count wanted
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    assert np.array_equal(out, ref)
    # the ctx's ordinary device calls (with their count workspace) are unaffected by the pipeline's own streams
    d = eng.to_device(verts)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.sat_rect_pairs_verts([d.row(k) for k in range(16)], n, d_out, d_cnt)
    assert int(d_cnt.get()[0]) == ref_cnt and np.array_equal(d_out.get(), ref)
    for a in (d, d_out, d_cnt):
        a.free()
