"""Graph-capture check, run as a separate process by tests/test_gpu_graph.py.

torch must be imported before libc2d.so in a process that uses both: torch bundles its own
libamdhip64.so.7, and whichever copy is loaded first serves the whole process (bench.py has the
same import order).  TEST INFRASTRUCTURE: uses the oracle as the checker."""
import os
import sys

import torch  # noqa: F401  (first: see above)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402

eng = pkg.Engine(0)


def main():
    dev = torch.device("cuda", 0)
    n = 50_000
    poses_h = wl.random_obb_pose_planes(n, seed=21, extent=3.0)
    pose = torch.from_numpy(poses_h).to(dev)
    planes = torch.zeros((16, n), dtype=torch.float32, device=dev)
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    hits = torch.zeros(1, dtype=torch.int64, device=dev)
    sc = wl.MC_PAIR_SCENE
    # adaptive scenes: tables + scenes + outputs allocated before capture; one eager call sizes the ctx workspace
    tp, ts, _ = wl.random_tables(32, 32, seed=9)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    ns = 3000
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 32, d_s, 32, 4.07, 1.74, 4.0, 3, 0, ns, d_sc)
    d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
    eng.mc_scenes(d_p, 32, d_s, 32, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 2000, 5, 0, d_h, d_u, None)
    ref_h, ref_u = d_h.get(), d_u.get()
    eng.memset(d_h, 0, ns * 4)
    eng.memset(d_u, 0, ns * 4)
    eng.synchronize()
    row = lambda t, k: t.data_ptr() + k * t.stride(0) * t.element_size()  # noqa: E731

    lib = eng.lib
    import ctypes as C

    def mc_scenes_async(stream):
        # the async form: no host outputs requested
        from c2d_amd import binding

        bins = np.asarray(wl.DEFAULT_BINS, np.float32)
        acc = np.asarray(wl.DEFAULT_BIN_ACCURACY, np.float32)
        a = binding._McScenesArgs(d_p.ptr, 32, d_s.ptr, 32, d_sc.ptr, ns, 4.07, 1.74, bins.ctypes.data_as(C.POINTER(C.c_float)),
                                  acc.ctypes.data_as(C.POINTER(C.c_float)), 4, 2000, 5, 0, 0, 0, 0, d_h.ptr, d_u.ptr, None, None, None)
        assert lib.c2d_mc_scenes(eng.h, C.byref(a), C.c_void_p(stream)) == 0

    # polygons (config 5 entry point): validation happens inside the kernel, so the call is capturable
    npoly = 20_000
    pvx_h, pvy_h, pk_h = wl.random_convex_polygons(npoly, seed=77, extent=3.0)
    pvx, pvy, pk = (torch.from_numpy(x).to(dev) for x in (pvx_h, pvy_h, pk_h))
    pout = torch.zeros(npoly, dtype=torch.uint8, device=dev)
    pcnt = torch.zeros(1, dtype=torch.int64, device=dev)
    pose_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)

    # the same polygons as a binned batch (made before the capture: building the table synchronises); test + results are capturable
    bins = eng.poly_bins_from_padded(pvx.data_ptr(), pvy.data_ptr(), pk.data_ptr(), npoly, 16, 2)
    bout = torch.zeros(npoly, dtype=torch.uint8, device=dev)
    bcnt = torch.zeros(1, dtype=torch.int64, device=dev)

    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=side):
        sh = torch.cuda.current_stream(dev).cuda_stream
        for r in range(2):
            eng.rects_from_poses(*[row(pose, 5 * r + k) for k in range(5)], n, [row(planes, 8 * r + k) for k in range(8)], stream=sh)
        eng.sat_rect_pairs_verts([row(planes, k) for k in range(16)], n, out.data_ptr(), cnt.data_ptr(), stream=sh)
        eng.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1, 2, 3, 100_000, hits.data_ptr(), stream=sh)
        mc_scenes_async(sh)
        eng.sat_poly_pairs(pvx.data_ptr(), pvy.data_ptr(), pk.data_ptr(), npoly, pout.data_ptr(), pcnt.data_ptr(), stream=sh)
        eng.sat_rect_pairs_pose([row(pose, k) for k in range(10)], n, pose_out.data_ptr(), None, stream=sh)
        eng.sat_rect_pairs_verts_mask([row(planes, k) for k in range(16)], n, mask.data_ptr(), None, stream=sh)
        eng.sat_poly_pairs_binned(bins, bcnt.data_ptr(), stream=sh)
        bins.results(bout.data_ptr(), stream=sh)
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0 and not out.any(), "capture must not execute anything"
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ref_planes = np.concatenate([oracle.rects_from_poses(*poses_h[:5]), oracle.rects_from_poses(*poses_h[5:])])
    ref_out, ref_cnt = oracle.sat_rect_pairs_verts(ref_planes)
    assert np.array_equal(planes.cpu().numpy().view(np.uint32), ref_planes.view(np.uint32))
    assert np.array_equal(out.cpu().numpy(), ref_out)
    assert int(cnt.item()) == 3 * ref_cnt                       # the counter accumulates over the three replays
    ref_hits = oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1, 2, 3, 100_000)
    assert int(hits.item()) == 3 * ref_hits
    assert np.array_equal(d_h.get(), ref_h) and np.array_equal(d_u.get(), ref_u)   # scenes re-zero their counters per call
    ref_poly, ref_pcnt = oracle.sat_poly_pairs(pvx_h, pvy_h, pk_h)
    assert np.array_equal(pout.cpu().numpy(), ref_poly) and int(pcnt.item()) == 3 * ref_pcnt
    assert np.array_equal(bout.cpu().numpy(), ref_poly) and int(bcnt.item()) == 3 * ref_pcnt
    assert np.array_equal(pose_out.cpu().numpy(), ref_out)
    assert np.array_equal(np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n], ref_out)
    eng.check_async()


if __name__ == "__main__":
    main()
    print("graph capture ok")
