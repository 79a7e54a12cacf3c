"""GPU tests: nothing writes outside its output.  Every output of every device entry point is placed between two 64 KiB guard
bands of a known byte inside one allocation, at an offset that keeps only the alignment include/c2d.h asks for; after the call
the bands must be untouched and the output must equal that of the same call into an ordinary buffer.  Sizes sit on and around the
kernels' wave, block and tile edges.  (What a kernel READS beyond its inputs cannot be seen this way: for the binning pass's move
kernel, whose hand-made indexing earned it, there is an index-checked build — tests/test_gpu_poly_binned.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BAND = 64 * 1024
FILL = 0xA5
SIZES = [1, 63, 64, 65, 255, 257, 1023, 4097, 8191, 8193, 70_001]


class Banded:
    """`nbytes` bytes of device memory at `ptr`, `misalign` bytes past a 256-byte boundary, with a guard band on either side"""

    def __init__(self, eng, nbytes, misalign=0):
        self.eng, self.nbytes = eng, nbytes
        self.total = BAND + 256 + nbytes + BAND
        self.base = eng.malloc(self.total)
        eng.memset(self.base, FILL, self.total)
        self.off = BAND + misalign
        self.ptr = self.base + self.off

    def get(self, dtype):
        """the payload; raises if either band was written"""
        raw = self.eng.read(self.base, (self.total,), np.uint8)
        before, after = raw[:self.off], raw[self.off + self.nbytes:]
        assert (before == FILL).all(), f"wrote {int((before != FILL).sum())} bytes BEFORE the output (nearest {self.off - int(np.flatnonzero(before != FILL)[-1])} bytes in front)"
        assert (after == FILL).all(), f"wrote {int((after != FILL).sum())} bytes AFTER the output (furthest {int(np.flatnonzero(after != FILL)[-1]) + 1} bytes behind)"
        return raw[self.off:self.off + self.nbytes].view(dtype)

    def free(self):
        self.eng.free(self.base)


def pose_pairs(eng, oracle, wl, n, seed):
    poses = wl.random_obb_pose_planes(n, seed=seed, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    return poses, planes


@pytest.mark.parametrize("n", SIZES)
def test_rectangle_pair_outputs(eng, oracle, wl, n):
    poses, planes = pose_pairs(eng, oracle, wl, n, 900 + n)
    ref, ref_cnt = oracle.sat_rect_pairs_verts(planes)
    d_pl, d_po = eng.to_device(planes), eng.to_device(poses)
    aos = np.ascontiguousarray(planes.reshape(2, 8, n).transpose(0, 2, 1))   # [rectangle][pair][8]
    d_aos = eng.to_device(aos)
    rows, prow = [d_pl.row(k) for k in range(16)], [d_po.row(k) for k in range(10)]
    for misalign in (0, 1):   # 16-byte aligned output: the wide-store instances; odd address: the plain ones
        for what, call in [("verts", lambda out, cnt: eng.sat_rect_pairs_verts(rows, n, out, cnt)),
                           ("aos", lambda out, cnt: eng.sat_rect_pairs_aos(d_aos.ptr, d_aos.ptr + 32 * n, n, out, cnt)),
                           ("pose", lambda out, cnt: eng.sat_rect_pairs_pose(prow, n, out, cnt))]:
            out, cnt = Banded(eng, n, misalign), Banded(eng, 8)
            eng.memset(cnt.ptr, 0, 8)
            call(out.ptr, cnt.ptr)
            eng.synchronize()
            assert np.array_equal(out.get(np.uint8), ref), (what, misalign)
            assert int(cnt.get(np.uint64)[0]) == ref_cnt, (what, misalign)
            out.free()
            cnt.free()
    words = (n + 63) // 64
    mask = Banded(eng, 8 * words)
    eng.sat_rect_pairs_verts_mask(rows, n, mask.ptr, None)
    eng.synchronize()
    bits = np.unpackbits(mask.get(np.uint64).view(np.uint8), bitorder="little")
    assert np.array_equal(bits[:n], ref) and not bits[n:].any()
    mask.free()
    # the rectangles themselves: eight planes, each in bands of its own
    outs = [Banded(eng, 4 * n, 4 * (k % 4)) for k in range(8)]
    eng.rects_from_poses(*prow[:5], n, [o.ptr for o in outs])
    eng.synchronize()
    for k, o in enumerate(outs):
        assert np.array_equal(o.get(np.float32).view(np.uint32), planes[k].view(np.uint32)), k
        o.free()
    for a in (d_pl, d_po, d_aos):
        a.free()


@pytest.mark.parametrize("rows", [4, 8, 13, 16])
@pytest.mark.parametrize("n", [1, 64, 65, 4097, 8193, 20_000])
def test_polygon_pair_outputs(eng, oracle, wl, n, rows):
    vx, vy, k = wl.random_convex_polygons(n, seed=31 * n + rows, kmin=1, kmax=rows, extent=1.5, rows=rows)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    for misalign in (0, 3):
        out, cnt = Banded(eng, n, misalign), Banded(eng, 8)
        eng.memset(cnt.ptr, 0, 8)
        eng.sat_poly_pairs_rows(dvx, dvy, dk, n, rows, out.ptr, cnt.ptr)
        eng.synchronize()
        assert np.array_equal(out.get(np.uint8), ref) and int(cnt.get(np.uint64)[0]) == ref_cnt, misalign
        out.free()
        cnt.free()
    # through the device binning: the results come back in the padded order, into the caller's buffer
    for g in (1, 3):
        bins = eng.poly_bins_from_padded(dvx, dvy, dk, n, rows, g)
        eng.sat_poly_pairs_binned(bins, None)
        out = Banded(eng, n, 1)
        bins.results(out.ptr)
        eng.synchronize()
        assert np.array_equal(out.get(np.uint8), ref), g
        out.free()
        bins.close()
    for a in (dvx, dvy, dk):
        a.free()


@pytest.mark.parametrize("n", [1, 63, 65, 1000, 4097])
def test_monte_carlo_scene_outputs(eng, pkg, wl, n):
    W, H, bins, acc = 4.07, 1.74, (0.0, 0.01, 0.1, 1.0), (1e-4, 1e-3, 1e-2)
    tp, ts, _ = wl.random_tables(64, 64, seed=n)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    sc = Banded(eng, n * pkg.SCENE_DT.itemsize)
    eng.sample_scenes(d_p, 64, d_s, 64, W, H, 4.0, 5, 0, n, sc.ptr)
    eng.synchronize()
    plain_sc = eng.empty(n, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 64, d_s, 64, W, H, 4.0, 5, 0, n, plain_sc)
    scenes = sc.get(np.uint8).copy()
    assert np.array_equal(scenes, plain_sc.get().view(np.uint8))
    # the adaptive loop: hits, samples used and the reference's five-float rows
    d_h, d_u, d_r = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32), eng.empty(n, pkg.ROW_DT)
    total, _ = eng.mc_scenes(d_p, 64, d_s, 64, plain_sc, n, W, H, bins, acc, 20_000, 11, 0, d_h, d_u, d_r)
    h, u, r = Banded(eng, 4 * n), Banded(eng, 4 * n), Banded(eng, n * pkg.ROW_DT.itemsize)
    eng.memset(h.ptr, 0, 4 * n)
    eng.memset(u.ptr, 0, 4 * n)
    total2, _ = eng.mc_scenes(d_p, 64, d_s, 64, plain_sc, n, W, H, bins, acc, 20_000, 11, 0, h.ptr, u.ptr, r.ptr)
    assert total2 == total
    assert np.array_equal(h.get(np.uint32), d_h.get()) and np.array_equal(u.get(np.uint32), d_u.get())
    assert np.array_equal(r.get(np.uint8), d_r.get().view(np.uint8))
    for a in (sc, h, u, r):
        a.free()
    for a in (d_p, d_s, plain_sc, d_h, d_u, d_r):
        a.free()


@pytest.mark.parametrize("n", [1, 65, 700])
def test_polygon_monte_carlo_scene_outputs(eng, pkg, wl, n):
    bins, acc = (0.0, 0.01, 0.1, 1.0), (1e-4, 1e-3, 1e-2)
    poses, sds = wl.random_poly_tables(40, 40, seed=4 + n)
    robot = wl.mc_poly_pair_scene(6, 5, seed=8)["robot"]
    scenes = wl.random_poly_scenes(n, poses, sds, 2.3, seed=9)
    d_p, d_s, d_sc = eng.to_device(poses), eng.to_device(sds), eng.to_device(scenes)
    d_h, d_u, d_r = eng.zeros(n, np.uint32), eng.zeros(n, np.uint32), eng.empty(n, pkg.ROW_DT)
    total, _ = eng.mc_poly_scenes(robot, d_p, 40, d_s, 40, d_sc, n, bins, acc, 6000, 2, 0, d_h, d_u, d_r)
    h, u, r = Banded(eng, 4 * n), Banded(eng, 4 * n), Banded(eng, n * pkg.ROW_DT.itemsize)
    eng.memset(h.ptr, 0, 4 * n)
    eng.memset(u.ptr, 0, 4 * n)
    total2, _ = eng.mc_poly_scenes(robot, d_p, 40, d_s, 40, d_sc, n, bins, acc, 6000, 2, 0, h.ptr, u.ptr, r.ptr)
    assert total2 == total
    assert np.array_equal(h.get(np.uint32), d_h.get()) and np.array_equal(u.get(np.uint32), d_u.get())
    assert np.array_equal(r.get(np.uint8), d_r.get().view(np.uint8))
    for a in (h, u, r):
        a.free()
    for a in (d_p, d_s, d_sc, d_h, d_u, d_r):
        a.free()


@pytest.mark.parametrize("rows,dims", [(1, 1), (63, 3), (1000, 3), (4097, 5), (65_537, 2)])
def test_table_outputs(eng, rows, dims):
    lo, hi = np.linspace(-1.0, 0.0, dims, dtype=np.float32), np.linspace(1.0, 3.0, dims, dtype=np.float32)
    plain = eng.empty((rows, dims), np.float32)
    eng.uniform_table_minstd(plain, rows, dims, lo, hi, 17)
    t = Banded(eng, 4 * rows * dims, 4)
    eng.uniform_table_minstd(t.ptr, rows, dims, lo, hi, 17)
    eng.synchronize()
    table = t.get(np.float32).copy()
    assert np.array_equal(table.view(np.uint32), plain.get().reshape(-1).view(np.uint32))
    # the square root in place and out of place
    mags = eng.to_device(np.abs(table))
    s = Banded(eng, 4 * rows * dims, 8)
    eng.sqrt_f32(mags, s.ptr, rows * dims)
    eng.synchronize()
    assert np.array_equal(s.get(np.float32), np.sqrt(np.abs(table)))
    for a in (t, s):
        a.free()
    plain.free()
    mags.free()


@pytest.mark.parametrize("n,begin", [(1, 0), (5, 3), (64, 10**12 + 1), (1027, 2)])
def test_sample_stream_outputs(eng, n, begin):
    plain_n, plain_r = eng.empty((n, 5), np.float32), eng.empty((n, 6), np.uint32)
    eng.philox_normals(99, 12345, begin, n, plain_n, plain_r)
    nb, rb = Banded(eng, 20 * n, 4), Banded(eng, 24 * n, 4)
    eng.philox_normals(99, 12345, begin, n, nb.ptr, rb.ptr)
    eng.synchronize()
    assert np.array_equal(nb.get(np.uint32), plain_n.get().reshape(-1).view(np.uint32))
    assert np.array_equal(rb.get(np.uint32), plain_r.get().reshape(-1))
    for a in (nb, rb):
        a.free()
    plain_n.free()
    plain_r.free()


def test_device_info_never_writes_past_the_callers_struct(eng, pkg):
    """c2d_device_info grew at its end in 0.5 (pci_bus_id) and c2d_ctx_info cleared sizeof(the NEW struct) bytes of the
    caller's: a binary built against the 0.4 header was overrun by 32 bytes.  Since 0.6 the struct is filled through
    c2d_ctx_info_sized, which is told the caller's size; the exported symbol c2d_ctx_info writes the 0.4 layout only."""
    import ctypes as C

    full = 128 + 64 + 4 * 4 + 8 + 32            # include/c2d.h: name, arch, four ints, hbm_bytes, pci_bus_id
    old = full - 32                             # C2D_DEVICE_INFO_BYTES_0_4
    lib = eng.lib
    ref = eng.info()
    assert C.sizeof(pkg.binding._DeviceInfo) == full
    for size in (1, 100, 128, old, old + 1, full - 1, full, full + 64):
        buf = (C.c_ubyte * (full + 256))(*([0xA5] * (full + 256)))
        rc = lib.c2d_ctx_info_sized(eng.h, C.cast(buf, C.POINTER(pkg.binding._DeviceInfo)), C.c_size_t(size))
        assert rc == 0
        raw = bytes(buf)
        wrote = min(size, full)
        assert set(raw[wrote:]) == {0xA5}, size                      # nothing beyond the caller's size (nor beyond the struct's)
        assert raw[:min(wrote, len(ref["name"]))] == ref["name"].encode()[:min(wrote, len(ref["name"]))]
        if size >= full:
            assert raw[old:full].split(b"\0")[0].decode() == ref["pci_bus_id"]
    # the exported symbol of the 0.4 header: the old layout, not one byte more
    buf = (C.c_ubyte * (full + 256))(*([0xA5] * (full + 256)))
    assert lib.c2d_ctx_info(eng.h, C.cast(buf, C.POINTER(pkg.binding._DeviceInfo))) == 0
    raw = bytes(buf)
    assert set(raw[old:]) == {0xA5} and raw[:old].split(b"\0")[0].decode() == ref["name"]
    assert int.from_bytes(raw[old - 8:old], "little") == ref["hbm_bytes"]
    assert lib.c2d_ctx_info_sized(eng.h, C.cast(buf, C.POINTER(pkg.binding._DeviceInfo)), C.c_size_t(0)) == -1
