"""A new stream at a destroyed stream's address is ANOTHER stream (csrc/c2d_internal.hpp workspace_same_stream), run as
a separate process by tests/test_gpu_workspace_guard.py:   stream_alias_check.py rocm | pytorch

  rocm     libc2d.so binds ROCm's libamdhip64.so.7 (hipStreamGetId exists: the stream's number decides);
  pytorch  torch is imported first, so the process holds the libamdhip64.so.7 PyTorch ships (HIP 7.0 in this image: no
           hipStreamGetId; the address is all the guard has, and c2d_stream_destroy forgets the address of a stream it
           destroys with tickets outstanding — include/c2d.h names what stays open for streams destroyed by other means).

An adaptive call (milliseconds of work) is queued on a stream, the stream destroyed — through the runtime, behind c2d's back,
where the runtime numbers its streams; through c2d_stream_destroy where it does not — new streams created until the address
repeats, and a counted call issued on that alias: it must be refused while the adaptive call is
certainly still running, give the oracle's count once the device has drained, and never disturb the adaptive call's rows.
TEST INFRASTRUCTURE: uses the oracle as the checker."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "rocm"
if mode == "pytorch":
    import torch  # noqa: F401  (loads PyTorch's libamdhip64 before libc2d.so is opened)

    torch.cuda.init()
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402


def main():
    eng = pkg.Engine(0)
    hip = C.CDLL("libamdhip64.so.7")   # the copy the process already holds, by soname
    has_id = hasattr(hip, "hipStreamGetId")
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def stream():
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        return s.value

    def counted(d, n, d_out, d_cnt, s):
        eng.memset(d_cnt.ptr, 0, 8, s)
        eng.sat_rect_pairs_verts([d.row(k) for k in range(16)], n, d_out, d_cnt, stream=s)

    tp, ts, _ = wl.random_tables(64, 64, seed=2)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    ns = 200_000
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, 4.07, 1.74, 4.0, 1, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    n = 50_001
    poses = wl.random_obb_pose_planes(n, seed=47)
    verts = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(verts)
    d = eng.to_device(verts)
    d_out, d_cnt = eng.zeros(n, np.uint8), eng.zeros(1, np.uint64)
    eng.synchronize()
    aliased, refused_on_alias, early, hits, used = 0, 0, 0, None, None
    for attempt in range(6):
        sa = stream()
        t0 = time.perf_counter()
        eng.mc_scenes_async(d_p, 64, d_s, 64, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 400_000, 3, 0, d_h, d_u, stream=sa)
        t_queued = time.perf_counter() - t0
        if has_id:
            assert hip.hipStreamDestroy(C.c_void_p(sa)) == 0  # behind c2d's back, with the adaptive call in flight
        else:
            eng.stream_destroy(sa)                             # (no stream numbers: c2d must be told, include/c2d.h)
        t_destroyed = time.perf_counter() - t0
        others, alias = [], None
        for _ in range(256):                                   # create streams until the address repeats
            t = stream()
            if t == sa:
                alias = t
                break
            others.append(t)
        t_alias, was_refused = None, False
        if alias is not None:
            aliased += 1
            t_alias = time.perf_counter() - t0
            try:
                counted(d, n, d_out, d_cnt, alias)             # same address, another stream: must not pass unchecked
            except pkg.C2DError as e:
                assert e.status == -5, e
                was_refused = True
                refused_on_alias += 1
        assert hip.hipDeviceSynchronize() == 0
        t_all = time.perf_counter() - t0
        print(f"  attempt {attempt}: call queued after {t_queued * 1e3:.2f} ms, stream destroyed after {t_destroyed * 1e3:.2f} ms, address repeated after "
              f"{len(others)} other streams at {(t_alias or 0) * 1e3:.2f} ms, alias call {'refused' if was_refused else 'accepted'}, device drained at {t_all * 1e3:.2f} ms")
        if t_alias is not None and t_alias < 0.5 * t_all:      # the adaptive call was certainly still running when the alias call came
            early += 1
            assert was_refused, ("a counted call on the alias passed while the adaptive call was running", attempt, t_alias, t_all, has_id)
        s = alias if alias is not None else others[0]
        counted(d, n, d_out, d_cnt, s)
        assert hip.hipStreamSynchronize(C.c_void_p(s)) == 0
        assert int(eng.read(d_cnt.ptr, (1,), np.uint64, stream=s)[0]) == ref_cnt and np.array_equal(d_out.get(stream=s), ref), attempt
        if hits is None:
            hits, used = d_h.get(stream=s), d_u.get(stream=s)
        else:                                                  # the adaptive call was never disturbed by the call on the alias
            assert np.array_equal(d_h.get(stream=s), hits) and np.array_equal(d_u.get(stream=s), used), attempt
        for t in others + ([alias] if alias is not None else []):
            assert hip.hipStreamDestroy(C.c_void_p(t)) == 0
    # (if the runtime never reuses the address, or the adaptive call retires before the address repeats, the hazard did not arise)
    print(f"alias check ok ({mode}: hipStreamGetId {'present' if has_id else 'absent'}): address repeated in {aliased} of 6 attempts, "
          f"{early} of them with the adaptive call certainly in flight, {refused_on_alias} alias calls refused")
    for a in (d_p, d_s, d_sc, d_h, d_u, d, d_out, d_cnt):
        a.free()
    eng.close()


if __name__ == "__main__":
    main()
