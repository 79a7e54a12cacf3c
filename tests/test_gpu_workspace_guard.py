"""The ctx workspace guard against streams whose lifetime belongs to the CALLER (include/c2d.h "streams"; the reference runs
everything on the default stream, compute_collision_probability.cu:288-310, and has no such hazard).

Round 4's guard asked the runtime about the previous call's stream (hipStreamQuery); a stream the caller had destroyed
meanwhile made that a query on a dangling handle (profiles/notes_r05_workspace_guard.md).  The guard now reads completion
stamps that the kernels raise themselves (csrc/c2d_internal.hpp) and never hands a remembered stream to the runtime.  These tests
create and destroy streams BEHIND c2d's back — through the HIP runtime directly, as a PyTorch or C++ caller would."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HIP_STREAM_NON_BLOCKING = 1


@pytest.fixture(scope="module")
def hip():
    lib = C.CDLL("libamdhip64.so")
    lib.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lib.hipStreamDestroy.argtypes = [C.c_void_p]
    lib.hipStreamSynchronize.argtypes = [C.c_void_p]
    lib.hipDeviceSynchronize.argtypes = []

    class Hip:
        def stream(self) -> int:
            s = C.c_void_p()
            assert lib.hipStreamCreateWithFlags(C.byref(s), HIP_STREAM_NON_BLOCKING) == 0
            return s.value

        def destroy(self, s: int):
            assert lib.hipStreamDestroy(C.c_void_p(s)) == 0

        def sync(self, s: int):
            assert lib.hipStreamSynchronize(C.c_void_p(s)) == 0

        def device_sync(self):
            assert lib.hipDeviceSynchronize() == 0

    return Hip()


def _pairs(eng, oracle, wl, n, seed):
    poses = wl.random_obb_pose_planes(n, seed=seed)
    verts = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    return eng.to_device(verts), ref, ref_cnt


def _counted(eng, d, n, d_out, d_cnt, stream):
    eng.memset(d_cnt.ptr, 0, 8, stream)
    eng.sat_rect_pairs_verts([d.row(k) for k in range(16)], n, d_out, d_cnt, stream=stream)


def test_counted_calls_survive_streams_destroyed_behind_the_ctx(eng, hip, oracle, wl):
    """Counted call on a foreign stream; the caller synchronises and destroys it with the runtime's own calls (c2d never hears
    of it); the next counted call, on another stream, must neither touch the dead handle nor lose a count.  Repeated over 40
    generations of streams so that the runtime reuses addresses."""
    n = 100_003
    d, ref, ref_cnt = _pairs(eng, oracle, wl, n, seed=41)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.synchronize()
    for gen in range(40):
        s = hip.stream()
        _counted(eng, d, n, d_out, d_cnt, s)       # a new stream: the guard reads stamps, it does not query the dead `prev`
        hip.sync(s)
        assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=s)[0]) == ref_cnt, gen
        if gen % 7 == 0:
            assert np.array_equal(d_out.get(stream=s), ref)
        hip.destroy(s)                              # not c2d_stream_destroy
    # and the default stream right after a destroyed one
    _counted(eng, d, n, d_out, d_cnt, 0)
    eng.synchronize()
    assert int(d_cnt.get()[0]) == ref_cnt
    for a in (d, d_out, d_cnt):
        a.free()


def test_stream_destroyed_with_work_in_flight(eng, pkg, hip, oracle, wl):
    """A long adaptive call queued on a foreign stream, the stream destroyed at once (the runtime may or may not drain it
    first): a counted call on another stream is then either refused (work in flight) or accepted (drained) — never a fault,
    never a wrong count — and is accepted once the device has drained."""
    tp, ts, _ = wl.random_tables(64, 64, seed=2)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    ns = 200_000
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, 4.07, 1.74, 4.0, 1, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    n = 50_001
    d, ref, ref_cnt = _pairs(eng, oracle, wl, n, seed=43)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.synchronize()
    sa, sb = hip.stream(), hip.stream()
    eng.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 400_000, 3, 0, d_h, d_u, stream=sa)
    hip.destroy(sa)
    try:
        _counted(eng, d, n, d_out, d_cnt, sb)
        refused = False
    except pkg.C2DError as e:
        assert e.status == -5
        refused = True
    hip.device_sync()
    _counted(eng, d, n, d_out, d_cnt, sb)
    hip.sync(sb)
    assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=sb)[0]) == ref_cnt and np.array_equal(d_out.get(stream=sb), ref), refused
    # the adaptive call itself was not disturbed: the same call again on a live stream gives the same rows
    hits, used = d_h.get(stream=sb), d_u.get(stream=sb)
    d_h2, d_u2 = eng.zeros(ns, np.uint32, stream=sb), eng.zeros(ns, np.uint32, stream=sb)
    eng.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 400_000, 3, 0, d_h2, d_u2, stream=sb)
    hip.sync(sb)
    assert np.array_equal(d_h2.get(stream=sb), hits) and np.array_equal(d_u2.get(stream=sb), used)
    hip.destroy(sb)
    for a in (d_p, d_s, d_sc, d_h, d_u, d_h2, d_u2, d, d_out, d_cnt):
        a.free()


@pytest.mark.parametrize("runtime", ["rocm", "pytorch"])
def test_a_new_stream_at_a_destroyed_streams_address_is_another_stream(runtime):
    """the guard lets a call pass unchecked when it comes on the stream the outstanding tickets were issued on — and
    compared ADDRESSES to decide that.  A stream destroyed with work in flight and a new one the runtime creates in its place
    (profiles/r05_stream_lifetime_probe.txt, cases 2 and 4: one of the next 64 streams lands there) are two streams: nothing
    orders the new one's calls behind the old one's.  tests/stream_alias_check.py builds exactly that, in a process of its own,
    against both HIP runtimes a caller may hold: ROCm's (hipStreamGetId decides) and the one PyTorch ships and loads first (HIP
    7.0: no such call, the idle query of the live stream and the stamps decide)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "stream_alias_check.py"), runtime], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "alias check ok" in out.stdout, out.stdout
    print(out.stdout.strip().splitlines()[-1])


def test_a_host_batch_does_not_disarm_the_guard_of_another_stream(eng, pkg, hip, oracle, wl):
    """Round 4 cleared the guard at the end of every host-batch call, also when that call never owned the workspace (no count
    wanted): a counted call on a third stream then passed unchecked while the first stream's call was still running."""
    tp, ts, _ = wl.random_tables(64, 64, seed=2)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    ns = 400_000
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, 4.07, 1.74, 4.0, 1, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    n = 5000
    poses = wl.random_obb_pose_planes(n, seed=9)
    verts = np.ascontiguousarray(np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])]))
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    d = eng.to_device(verts)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.synchronize()
    sa, sc = hip.stream(), hip.stream()
    eng.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 1_000_000, 3, 0, d_h, d_u,
                        stream=sa)   # a hundred milliseconds of queued work on stream A
    arr = (C.c_void_p * 16)(*[verts[k].ctypes.data for k in range(16)])
    out = np.zeros(n, np.uint8)
    assert eng.lib.c2d_sat_rect_pairs_verts_host(eng.h, arr, n, C.c_void_p(out.ctypes.data), None) == 0   # no count: no workspace
    assert np.array_equal(out, ref)
    with pytest.raises(pkg.C2DError) as ei:
        _counted(eng, d, n, d_out, d_cnt, sc)       # stream A is still busy: refused
    assert ei.value.status == -5
    hip.sync(sa)                                    # behind c2d's back again
    _counted(eng, d, n, d_out, d_cnt, sc)
    hip.sync(sc)
    assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=sc)[0]) == ref_cnt
    # a host batch WITH its count, then a device call on a foreign stream
    cnt = eng.sat_rect_pairs_host([verts[k] for k in range(16)], out, "verts")
    assert cnt == ref_cnt
    _counted(eng, d, n, d_out, d_cnt, sc)
    hip.sync(sc)
    assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=sc)[0]) == ref_cnt
    hip.destroy(sa)
    hip.destroy(sc)
    for a in (d_p, d_s, d_sc, d_h, d_u, d, d_out, d_cnt):
        a.free()


def test_every_counting_kernel_raises_its_stamps(eng, hip, oracle, wl):
    """Each counted entry point (single-level count: vertex, bit-mask, array-of-rectangles, pose, 4-row polygons; two-level
    count: padded and binned polygons) followed by a counted call on a different stream after a runtime-side synchronise: if
    one of them failed to raise a stamp the second call would be refused for ever."""
    n = 70_001
    poses = wl.random_obb_pose_planes(n, seed=77)
    verts = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    _, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    d, dp = eng.to_device(verts), eng.to_device(poses)
    d1, d2 = eng.to_device(np.ascontiguousarray(verts[:8].T)), eng.to_device(np.ascontiguousarray(verts[8:].T))
    d_out, d_cnt, d_mask = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64), eng.zeros((n + 63) // 64, np.uint64)
    vx, vy, k = wl.random_convex_polygons(n, seed=5)
    _, poly_cnt = oracle.sat_poly_pairs(vx, vy, k)
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    vx4, vy4, k4 = wl.random_convex_polygons(n - n % 4, seed=6, kmin=3, kmax=4, rows=4)
    _, poly4_cnt = oracle.sat_poly_pairs(vx4, vy4, k4)
    dvx4, dvy4, dk4 = eng.to_device(vx4), eng.to_device(vy4), eng.to_device(k4)
    eng.synchronize()
    bins = eng.poly_bins_from_padded(dvx, dvy, dk, n, 16, 1)
    calls = [
        (lambda s: eng.sat_rect_pairs_verts([d.row(i) for i in range(16)], n, d_out, d_cnt, stream=s), ref_cnt),
        (lambda s: eng.sat_rect_pairs_verts_mask([d.row(i) for i in range(16)], n, d_mask, d_cnt, stream=s), ref_cnt),
        (lambda s: eng.sat_rect_pairs_aos(d1, d2, n, d_out, d_cnt, stream=s), ref_cnt),
        (lambda s: eng.sat_rect_pairs_pose([dp.row(i) for i in range(10)], n, d_out, d_cnt, stream=s), ref_cnt),
        (lambda s: eng.sat_poly_pairs(dvx, dvy, dk, n, d_out, d_cnt, stream=s), poly_cnt),
        (lambda s: eng.sat_poly_pairs_rows(dvx4, dvy4, dk4, n - n % 4, 4, d_out, d_cnt, stream=s), poly4_cnt),
        (lambda s: eng.sat_poly_pairs_binned(bins, d_cnt, stream=s), poly_cnt),
    ]
    for i, (call, want) in enumerate(calls):
        sa, sb = hip.stream(), hip.stream()
        eng.memset(d_cnt.ptr, 0, 8, sa)
        call(sa)
        hip.sync(sa)
        assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=sa)[0]) == want, i
        hip.destroy(sa)
        _counted(eng, d, n, d_out, d_cnt, sb)      # refused here = call i left a stamp behind its ticket
        hip.sync(sb)
        assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=sb)[0]) == ref_cnt, i
        hip.destroy(sb)
    bins.close()
    for a in (d, dp, d1, d2, d_out, d_cnt, d_mask, dvx, dvy, dk, dvx4, dvy4, dk4):
        a.free()


def test_tickets_wrap_after_two_to_the_32_launches(pkg, hip, oracle, wl):
    """Tickets travel as 32 bits (a 64-bit kernel argument cost the 16-row polygon kernel two spilled registers).  When they run out
    the guard drains the device, zeroes stamps and expectations and starts again at 1.  The rehearsal build of the library (tests
    only) can set the counter: walk it through the wrap with counted calls on alternating foreign streams — every count right,
    a busy stream still refused afterwards."""
    import ctypes as C
    import os

    reh = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "convex-2d-gpu-collision-detection_amd", "lib-rehearsal", "libc2d.so")
    e = pkg.Engine(0, lib_path=reh)
    hook = e.lib.c2d_test_set_workspace_ticket
    hook.restype, hook.argtypes = C.c_int, [C.c_void_p, C.c_uint]
    n = 70_001
    d, ref, ref_cnt = _pairs(e, oracle, wl, n, seed=91)
    vx, vy, k = wl.random_convex_polygons(n, seed=9)
    _, poly_cnt = oracle.sat_poly_pairs(vx, vy, k)
    dvx, dvy, dk = e.to_device(vx), e.to_device(vy), e.to_device(k)
    d_out, d_cnt = e.zeros(n, np.uint8), e.zeros(1, np.uint64)
    e.synchronize()
    assert hook(e.h, 0xFFFFFFF8) == 0
    for i in range(24):      # tickets 0xfffffff9 .. 0xffffffff, the wrap, 1 .. : a new stream every time, so every call is checked
        s = hip.stream()
        e.memset(d_cnt.ptr, 0, 8, s)
        if i % 3 == 2:       # the two-level count
            e.sat_poly_pairs(dvx, dvy, dk, n, d_out, d_cnt, stream=s)
            want = poly_cnt
        else:
            e.sat_rect_pairs_verts([d.row(j) for j in range(16)], n, d_out, d_cnt, stream=s)
            want = ref_cnt
        hip.sync(s)
        assert int(e.read(d_cnt.ptr, (1,), np.uint64, stream=s)[0]) == want, i
        hip.destroy(s)
    # after the wrap the guard still guards
    tp, ts, _ = wl.random_tables(64, 64, seed=2)
    d_p, d_s = e.to_device(tp), e.to_device(ts)
    ns = 200_000
    d_sc = e.empty(ns, pkg.SCENE_DT)
    e.sample_scenes(d_p, 64, d_s, 64, 4.07, 1.74, 4.0, 1, 0, ns, d_sc)
    d_h, d_u = e.zeros(ns, np.uint32), e.zeros(ns, np.uint32)
    e.synchronize()
    sa, sb = hip.stream(), hip.stream()
    e.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 400_000, 3, 0, d_h, d_u, stream=sa)
    with pytest.raises(pkg.C2DError) as ei:
        _counted(e, d, n, d_out, d_cnt, sb)
    assert ei.value.status == -5
    hip.sync(sa)
    _counted(e, d, n, d_out, d_cnt, sb)
    hip.sync(sb)
    assert int(e.read(d_cnt.ptr, (1,), np.uint64, stream=sb)[0]) == ref_cnt
    hip.destroy(sa)
    hip.destroy(sb)
    for a in (d, dvx, dvy, dk, d_out, d_cnt, d_p, d_s, d_sc, d_h, d_u):
        a.free()
    e.close()
