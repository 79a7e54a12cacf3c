"""GPU parity tests of the binned polygon path (include/c2d.h "binned polygon batches") through the C-ABI: every boolean
of every bin equals the oracle's for the same polygons, whatever the bin sizes, row counts, strides and densities; the
device binning of a padded batch returns the padded entry point's results in the padded order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_from_padded(eng, vx, vy, k, rows, g):
    n = vx.shape[-1]
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    bins = eng.poly_bins_from_padded(dvx, dvy, dk, n, rows, g)
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_binned(bins, d_cnt)
    d_out = eng.zeros(n + 8, np.uint8)
    bins.results(d_out)
    out, cnt = d_out.get(), int(d_cnt.get()[0])
    assert not out[n:].any()
    for a in (dvx, dvy, dk, d_cnt, d_out):
        a.free()
    return out[:n], cnt, bins


@pytest.mark.parametrize("g", [1, 2, 3, 4, 8, 16])
@pytest.mark.parametrize("n,kmin,kmax,extent", [(1, 3, 16, 1.0), (63, 3, 16, 1.0), (200_001, 3, 16, 8.0), (50_000, 3, 16, 1.0), (30_000, 1, 16, 0.6)])
def test_from_padded_matches_oracle(eng, oracle, wl, g, n, kmin, kmax, extent):
    vx, vy, k = wl.random_convex_polygons(n, seed=17 * g + n, kmin=kmin, kmax=kmax, extent=extent)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    out, cnt, bins = run_from_padded(eng, vx, vy, k, 16, g)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    assert bins.pairs == n and len(bins) <= (16 // g + (16 % g != 0)) ** 2
    # what one test moves: the bins' own rows (counts only where a bin can hold different sizes)
    want = 0
    for i in range(len(bins)):
        b = bins.get(i)
        assert b["stride"] >= b["n"] and b["stride"] % 64 == 0 and (b["ka"] != 0) == (g > 1)
        want += b["n"] * ((b["rows_a"] + b["rows_b"]) * 8 + (2 if g > 1 else 0) + 1)
    assert bins.bytes == want
    if g == 1:
        assert bins.bytes == int(k.astype(np.int64).sum()) * 8 + n     # exactly the vertices + one result byte per pair
    bins.close()


@pytest.mark.parametrize("rows,g", [(4, 1), (8, 2), (12, 4), (5, 3), (1, 1)])
def test_from_padded_row_layouts(eng, oracle, wl, rows, g):
    n = 40_003
    vx, vy, k = wl.random_convex_polygons(n, seed=rows * 100 + g, kmin=min(3, rows), kmax=rows, extent=1.2, rows=rows)
    ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
    out, cnt, bins = run_from_padded(eng, vx, vy, k, rows, g)
    assert np.array_equal(out, ref) and cnt == ref_cnt
    for i in range(len(bins)):
        b = bins.get(i)
        assert b["rows_a"] <= rows and b["rows_b"] <= rows
    bins.close()


def _padded_of_bin(eng, b):
    """download one bin and lay it out as the oracle's padded arrays"""
    n, st = b["n"], b["stride"]
    vx, vy = np.zeros((2, 16, n), np.float32), np.zeros((2, 16, n), np.float32)
    k = np.empty((2, n), np.uint8)
    for p, (px, py, pk, rows) in enumerate(((b["ax"], b["ay"], b["ka"], b["rows_a"]), (b["bx"], b["by"], b["kb"], b["rows_b"]))):
        vx[p, :rows] = eng.read(px, (rows, st), np.float32)[:, :n]
        vy[p, :rows] = eng.read(py, (rows, st), np.float32)[:, :n]
        k[p] = eng.read(pk, (n,), np.uint8) if pk else rows
    return vx, vy, k


def test_every_bin_is_self_consistent(eng, oracle, wl):
    """The bins made on the device hold real polygons: the oracle on each bin's own planes gives that bin's result bytes,
    and the multiset of polygons is the input's (checked through the count of pairs per (ka, kb))."""
    n = 60_000
    vx, vy, k = wl.random_convex_polygons(n, seed=5, extent=1.5)
    for g in (1, 4):
        out, cnt, bins = run_from_padded(eng, vx, vy, k, 16, g)
        seen = np.zeros((17, 17), np.int64)
        total = 0
        for i in range(len(bins)):
            b = bins.get(i)
            bx, by, bk = _padded_of_bin(eng, b)
            assert (bk[0] <= b["rows_a"]).all() and (bk[1] <= b["rows_b"]).all()
            ref, ref_cnt = oracle.sat_poly_pairs(bx, by, bk)
            assert np.array_equal(eng.read(b["out"], (b["n"],), np.uint8), ref), (g, i)
            total += ref_cnt
            np.add.at(seen, (bk[0], bk[1]), 1)
        want = np.zeros((17, 17), np.int64)
        np.add.at(want, (k[0], k[1]), 1)
        assert np.array_equal(seen, want) and total == cnt
        bins.close()


def _upload_user_bins(eng, rng, specs, extent, wl, counted, stride_pad=0, seed=0):
    """specs: [(rows_a, rows_b, n)] -> (device-side bin dicts, host copies for the oracle, buffers to free)"""
    bins, host, bufs = [], [], []
    for j, (ra, rb, n) in enumerate(specs):
        vx, vy, k = wl.random_convex_polygons(n, seed=seed + 31 * j + n, kmin=1, kmax=16, extent=extent)
        if counted:
            k[0] = rng.integers(1, ra + 1, n)
            k[1] = rng.integers(1, rb + 1, n)
        else:
            k[0], k[1] = ra, rb
        # re-draw polygons with exactly those counts (the generator picks counts itself, so build per count)
        for p in range(2):
            for kk in np.unique(k[p]):
                sel = np.flatnonzero(k[p] == kk)
                gx, gy, _ = wl.random_convex_polygons(len(sel), seed=seed + 7 * j + int(kk) + p, kmin=int(kk), kmax=int(kk), extent=extent)
                vx[p][:, sel], vy[p][:, sel] = gx[p], gy[p]
        st = n + stride_pad
        planes = {}
        for name, src, rows in (("ax", vx[0], ra), ("ay", vy[0], ra), ("bx", vx[1], rb), ("by", vy[1], rb)):
            h = np.full((rows, st), np.nan, np.float32)      # the stride padding is never read as a vertex of a real pair
            h[:, :n] = src[:rows]
            planes[name] = eng.to_device(h)
        d = {"rows_a": ra, "rows_b": rb, "n": n, "stride": 0 if stride_pad == 0 else st, **planes, "out": eng.zeros(n + 4, np.uint8)}
        if counted:
            d["ka"], d["kb"] = eng.to_device(k[0].copy()), eng.to_device(k[1].copy())
        bins.append(d)
        host.append((vx, vy, k))
        bufs += [v for v in d.values() if hasattr(v, "free")]
    return bins, host, bufs


@pytest.mark.parametrize("counted", [False, True])
@pytest.mark.parametrize("extent,stride_pad", [(6.0, 0), (0.8, 0), (1.5, 13)])
def test_user_bins_of_every_shape(eng, oracle, wl, counted, extent, stride_pad):
    """Bins as a caller would hand them over: odd sizes, one to sixteen rows on either side, a bin smaller than a wave,
    an empty bin, exact bins without count arrays and counted bins, padded strides; sparse and dense scenes."""
    rng = np.random.default_rng(int(extent * 10) + stride_pad + counted)
    specs = [(16, 16, 4001), (3, 16, 700), (16, 3, 1300), (1, 1, 200), (2, 5, 333), (15, 16, 64), (8, 8, 5000), (9, 7, 129), (12, 4, 2),
             (4, 4, 0), (5, 11, 1), (10, 10, 1000), (13, 14, 65), (6, 6, 4096), (7, 2, 191)]
    bins, host, bufs = _upload_user_bins(eng, rng, specs, extent, wl, counted, stride_pad, seed=1000 * counted)
    handle = eng.poly_bins_create(bins)
    # the handle's bin i is the caller's bin i, empty ones included (they take no tiles): c2d_poly_bins_get never renumbers
    assert len(handle) == len(specs) and handle.pairs == sum(s[2] for s in specs)
    for i, (ra, rb, n) in enumerate(specs):
        g = handle.get(i)
        assert (g["rows_a"], g["rows_b"], g["n"]) == (ra, rb, n) and g["out"] == bins[i]["out"].ptr
    d_cnt = eng.zeros(1, np.uint64)
    eng.sat_poly_pairs_binned(handle, d_cnt)
    eng.sat_poly_pairs_binned(handle, None)            # the count is optional; results are rewritten identically
    total = 0
    for d, (vx, vy, k), (ra, rb, n) in zip(bins, host, specs):
        got = d["out"].get()
        assert not got[n:].any()
        if n == 0:
            continue
        ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
        assert np.array_equal(got[:n], ref), (ra, rb, n)
        total += ref_cnt
    assert int(d_cnt.get()[0]) == total
    eng.check_async()
    handle.close()
    for b in bufs + [d_cnt]:
        b.free()


def test_swapping_a_and_b_changes_nothing(eng, oracle, wl):
    rng = np.random.default_rng(4)
    specs = [(5, 12, 3000), (16, 3, 2000)]
    bins, host, bufs = _upload_user_bins(eng, rng, specs, 1.0, wl, True, seed=77)
    swapped = [dict(d, rows_a=d["rows_b"], rows_b=d["rows_a"], ax=d["bx"], ay=d["by"], bx=d["ax"], by=d["ay"], ka=d["kb"], kb=d["ka"],
                    out=eng.zeros(d["n"], np.uint8)) for d in bins]
    h1, h2 = eng.poly_bins_create(bins), eng.poly_bins_create(swapped)
    eng.sat_poly_pairs_binned(h1)
    eng.sat_poly_pairs_binned(h2)
    for d, s in zip(bins, swapped):
        assert np.array_equal(d["out"].get()[:d["n"]], s["out"].get())
    h1.close()
    h2.close()
    for b in bufs + [s["out"] for s in swapped]:
        b.free()


def test_non_finite_vertices_in_bins(eng, oracle, wl):
    n = 50_001
    vx, vy, k = wl.random_convex_polygons(n, seed=9, extent=1.5)
    bx = wl.inject_non_finite(vx.reshape(32, -1), seed=4).reshape(vx.shape)
    by = wl.inject_non_finite(vy.reshape(32, -1), seed=5, frac=0.2).reshape(vy.shape)
    ref, ref_cnt = oracle.sat_poly_pairs(bx, by, k)
    for g in (1, 4):
        out, cnt, bins = run_from_padded(eng, bx, by, k, 16, g)
        assert np.array_equal(out, ref) and cnt == ref_cnt
        bins.close()


def test_argument_errors(eng, pkg, wl):
    buf = eng.zeros(4096, np.float32)
    ok = {"rows_a": 4, "rows_b": 4, "n": 10, "ax": buf, "ay": buf, "bx": buf, "by": buf, "out": buf}
    for bad in ({"rows_a": 0}, {"rows_b": 17}, {"ax": None}, {"out": None}, {"ka": buf}, {"stride": 5}):
        with pytest.raises(pkg.C2DError):
            eng.poly_bins_create([dict(ok, **bad)])
    h = eng.poly_bins_create([])                     # no bins: a no-op
    eng.sat_poly_pairs_binned(h)
    assert len(h) == 0 and h.pairs == 0
    with pytest.raises(pkg.C2DError):
        h.results(buf)                               # not made by from_padded
    h.close()
    # a count outside 1..rows inside a counted bin: reported at the next synchronise, the pair reads 0
    kk = np.full(10, 3, np.uint8)
    kk[7] = 9
    dka, dkb = eng.to_device(kk), eng.to_device(np.full(10, 3, np.uint8))
    h = eng.poly_bins_create([dict(ok, ka=dka, kb=dkb, out=eng.zeros(16, np.uint8))])
    eng.sat_poly_pairs_binned(h)
    with pytest.raises(pkg.C2DError) as ei:
        eng.synchronize()
    assert ei.value.status == -1
    h.close()
    # from_padded with a bad count: refused with a message; an empty batch is fine
    vx, vy, k = wl.random_convex_polygons(100, seed=1)
    k[1, 37] = 17
    dvx, dvy, dk = eng.to_device(vx), eng.to_device(vy), eng.to_device(k)
    with pytest.raises(pkg.C2DError) as ei:
        eng.poly_bins_from_padded(dvx, dvy, dk, 100, 16, 1)
    assert "vertex count" in str(ei.value)
    for args in ((0, 16, 1), ):
        h = eng.poly_bins_from_padded(dvx, dvy, dk, *args)
        eng.sat_poly_pairs_binned(h)
        h.close()
    for rows, g in ((0, 1), (17, 1), (16, 0), (16, 17)):
        with pytest.raises(pkg.C2DError):
            eng.poly_bins_from_padded(dvx, dvy, dk, 100, rows, g)
    for a in (buf, dka, dkb, dvx, dvy, dk):
        a.free()


def test_binned_differential_fuzz(eng, oracle, wl):
    """40 random batches: random bin lists (rows, sizes, counted or exact, strides) and densities against the oracle."""
    rng = np.random.default_rng(2024)
    for it in range(40):
        nb = int(rng.integers(1, 12))
        specs = [(int(rng.integers(1, 17)), int(rng.integers(1, 17)), int(rng.choice([1, 5, 63, 64, 65, 300, 2000]))) for _ in range(nb)]
        counted = bool(rng.integers(0, 2))
        extent = float(rng.choice([0.5, 1.0, 2.0, 6.0]))
        bins, host, bufs = _upload_user_bins(eng, rng, specs, extent, wl, counted, int(rng.choice([0, 0, 3, 64])), seed=it * 100)
        h = eng.poly_bins_create(bins)
        d_cnt = eng.zeros(1, np.uint64)
        eng.sat_poly_pairs_binned(h, d_cnt)
        total = 0
        for d, (vx, vy, k), (ra, rb, n) in zip(bins, host, specs):
            ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
            assert np.array_equal(d["out"].get()[:n], ref), (it, ra, rb, n, counted, extent)
            total += ref_cnt
        assert int(d_cnt.get()[0]) == total
        h.close()
        for b in bufs + [d_cnt]:
            b.free()


@pytest.mark.parametrize("n,rows,g,kmin", [(20_000, 13, 1, 4), (63, 12, 11, 10), (4097, 16, 14, 10), (4096, 16, 2, 6), (8192, 9, 3, 2), (70_001, 7, 5, 7)])
def test_move_kernel_stays_inside_its_arrays(pkg, oracle, wl, capfd, n, rows, g, kmin):
    """The index-checked build of the binning pass (make lib-movecheck): every index poly_bin_move_kernel forms is compared with
    its array's size and the first offender is reported on stderr instead of being used.  A differential fuzz met a memory
    fault at the first shape: as hipcc 7.2 had compiled it, the kernel read its last, partial tile as a whole one, up to 8191
    floats past the end of vx / vy — values nobody used, so every result was right, and the reads only faulted when the batch
    ended where a mapping ended (profiles/notes_r05_move_kernel_overread.md).  The shapes: that one, partial tiles of one and of
    several tiles, a batch that is exactly one 4096-pair half tile and one that is exactly a tile."""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "convex-2d-gpu-collision-detection_amd", "lib", "libc2d_movecheck.so")
    e = pkg.Engine(0, lib_path=lib)
    try:
        vx, vy, k = wl.random_convex_polygons(n, seed=600_401 + n, kmin=kmin, kmax=rows, extent=1.5, rows=rows)
        ref, ref_cnt = oracle.sat_poly_pairs(vx, vy, k)
        capfd.readouterr()
        out, cnt, bins = run_from_padded(e, vx, vy, k, rows, g)
        err = capfd.readouterr().err
        bins.close()
    finally:
        e.close()
    assert "[c2d move check]" not in err, err
    assert np.array_equal(out, ref) and cnt == ref_cnt
