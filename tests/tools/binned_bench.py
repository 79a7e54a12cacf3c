#!/usr/bin/env python3
"""Developer tool: the BASELINE config-5 workload (K ~ U{kmin..kmax}, padded f32[2][16][n]) through the padded entry point
and through bins made from it at several granularities: time of the binning pass, time of one test over the bins
(back-to-back launches and isolated launches), bytes moved, results against the padded kernel's (all pairs) and against
the CPU oracle (a sample).  TEST INFRASTRUCTURE (uses oracle/).

usage: binned_bench.py [pairs] [reps] [kmin] [kmax] [extent] [granularities, e.g. 1,2,4]
C2D_LIBRARY=<other libc2d.so> selects another build of the same C-ABI for A/B runs."""
import os
import sys
import time

import torch  # before libc2d.so: one libamdhip64 per process
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import torch_random_convex_polygons  # noqa: E402

pkg = load_package()
from oracle import cpu as oracle  # noqa: E402


def timed(stream, fn, reps, isolated):
    """median ms per call: back-to-back (one event pair around all reps) or isolated (a synchronise between calls);
    150 ms of the same calls first, because the clocks ramp for ~15 ms after an idle period"""
    w0 = time.perf_counter()
    while time.perf_counter() - w0 < 0.15:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    if not isolated:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    kmin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    kmax = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    extent = float(sys.argv[5]) if len(sys.argv) > 5 else 8.0
    grans = [int(g) for g in (sys.argv[6] if len(sys.argv) > 6 else "1,2,4").split(",")]
    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    vx, vy, kk = torch_random_convex_polygons(torch, dev, n, seed=0xC0FFEE, kmin=kmin, kmax=kmax, extent=extent)
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    torch.cuda.synchronize()
    exact = int(kk.to(torch.int64).sum().item()) * 8 + 3 * n

    def padded():
        eng.sat_poly_pairs(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, out.data_ptr(), None, stream=sh)

    for _ in range(5):
        padded()
    torch.cuda.synchronize()
    ms = timed(stream, padded, reps, False)
    ms_iso = timed(stream, padded, reps, True)
    print(f"pairs {n} K~U{{{kmin}..{kmax}}} extent {extent}: exact bytes {exact / n:.1f} B/pair")
    print(f"padded      : {ms:.4f} ms back-to-back, {ms_iso:.4f} isolated; 259 B/pair -> {259 * n / ms / 1e6:.0f} GB/s; exact {exact / ms / 1e6:.0f} GB/s "
          f"({exact / ms / 1e6 / 8000:.3f} of 8 TB/s)")
    ref_out = out.clone()
    m = min(n, 300_000)
    ref, _ = oracle.sat_poly_pairs(vx[:, :, :m].contiguous().cpu().numpy(), vy[:, :, :m].contiguous().cpu().numpy(), kk[:, :m].contiguous().cpu().numpy())
    bad_total = int((ref != ref_out[:m].cpu().numpy()).sum())
    for g in grans:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bins = eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, 16, g, stream=sh)
        t_bin = (time.perf_counter() - t0) * 1e3
        bins2 = eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, 16, g, stream=sh)   # second time: allocator warm
        t0 = time.perf_counter()
        bins3 = eng.poly_bins_from_padded(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, 16, g, stream=sh)
        t_bin2 = (time.perf_counter() - t0) * 1e3
        bins2.close()
        bins3.close()
        cnt.zero_()

        def binned():
            eng.sat_poly_pairs_binned(bins, cnt.data_ptr(), stream=sh)

        def binned_nocount():
            eng.sat_poly_pairs_binned(bins, None, stream=sh)

        for _ in range(5):
            binned()
        torch.cuda.synchronize()
        cnt.zero_()
        torch.cuda.synchronize()
        binned()
        torch.cuda.synchronize()
        c_after = int(cnt.item())
        bms = timed(stream, binned, reps, False)
        bms_iso = timed(stream, binned, reps, True)
        bms_nc = timed(stream, binned_nocount, reps, False)
        got = torch.zeros(n, dtype=torch.uint8, device=dev)
        bins.results(got.data_ptr(), stream=sh)
        rms = timed(stream, lambda: bins.results(got.data_ptr(), stream=sh), reps, False)
        torch.cuda.synchronize()
        eng.check_async()
        diff = int((got != ref_out).sum().item())
        bad_total += diff
        moved = bins.bytes
        print(f"bins g={g:2d} ({len(bins):3d} bins): test {bms:.4f} ms back-to-back, {bms_iso:.4f} isolated, {bms_nc:.4f} without the count; moves {moved / n:.1f} B/pair -> "
              f"{moved / bms / 1e6:.0f} GB/s ({moved / bms / 1e6 / 8000:.3f}); exact {exact / bms / 1e6:.0f} GB/s ({exact / bms / 1e6 / 8000:.3f} of 8 TB/s); "
              f"binning pass {t_bin:.1f} ms first / {t_bin2:.1f} ms warm; results-to-input-order {rms:.4f} ms; "
              f"differs from the padded kernel on {diff} pairs; count ok {c_after == int(ref_out.sum().item())}")
        bins.close()
    print(f"oracle check on the first {m} pairs + binned-vs-padded on all: {bad_total} mismatches")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
