#!/usr/bin/env python3
"""Developer tool: time c2d_sat_poly_pairs on the BASELINE config-5 workload (HIP events on the
kernel's stream) and check a sample against the CPU oracle.  TEST INFRASTRUCTURE (uses oracle/).

usage: poly_bench.py [pairs] [reps] [kmin] [kmax] [extent] [sorted|rowsN]   ("rowsN": layout with N vertex rows per polygon,
c2d_sat_poly_pairs_rows)
       poly_bench.py [pairs] [reps] [kmin] [kmax] [extent] [sorted]   ("sorted": pairs ordered by (ka, kb), so that a
wave's pairs have equal vertex counts and the rows above them are skipped: traffic = the exact bytes)
C2D_LIBRARY=<other libc2d.so> selects another build of the same C-ABI for A/B runs."""
import os
import sys

import torch  # before libc2d.so: one libamdhip64 per process
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import torch_random_convex_polygons  # noqa: E402

pkg = load_package()
from oracle import cpu as oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    kmin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    kmax = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    extent = float(sys.argv[5]) if len(sys.argv) > 5 else 8.0
    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    vx, vy, kk = torch_random_convex_polygons(torch, dev, n, seed=0xC0FFEE, kmin=kmin, kmax=kmax, extent=extent)
    rows = 16
    if len(sys.argv) > 6 and sys.argv[6].startswith("rows"):
        rows = int(sys.argv[6][4:])
        vx, vy = vx[:, :rows, :].contiguous(), vy[:, :rows, :].contiguous()
    if len(sys.argv) > 6 and sys.argv[6] == "sorted":
        order = torch.argsort(kk[0].to(torch.int64) * 32 + kk[1].to(torch.int64))
        vx, vy, kk = vx[:, :, order].contiguous(), vy[:, :, order].contiguous(), kk[:, order].contiguous()
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    torch.cuda.synchronize()  # the inputs were generated on torch's current stream

    def step():
        eng.sat_poly_pairs_rows(vx.data_ptr(), vy.data_ptr(), kk.data_ptr(), n, rows, out.data_ptr(), cnt.data_ptr(), stream=sh)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    cnt.zero_()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        step()
        e1.record(stream)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    eng.check_async()
    ms = float(np.median(times))
    exact = int(kk.to(torch.int64).sum().item()) * 8 + 3 * n
    print(f"pairs {n} K~U{{{kmin}..{kmax}}} rows {rows} extent {extent}: median {ms:.4f} ms  min {min(times):.4f}  "
          f"{n / ms / 1e6:.3f} Gpairs/s  layout {(16 * rows + 3) * n / ms / 1e6:.0f} GB/s ({(16 * rows + 3) * n / ms / 1e6 / 8000:.3f} of 8 TB/s)  "
          f"exact {exact / ms / 1e6:.0f} GB/s  collide rate {cnt.item() / reps / n:.4f}")
    m = min(n, 300_000)
    sel = slice(n - m, n)
    ref, _ = oracle.sat_poly_pairs(vx[:, :, sel].contiguous().cpu().numpy(), vy[:, :, sel].contiguous().cpu().numpy(),
                                   kk[:, sel].contiguous().cpu().numpy())
    got = out[sel].cpu().numpy()
    bad = int((ref != got).sum())
    print(f"oracle check on the last {m} pairs: {bad} mismatches; count consistent: {int(cnt.item()) == reps * int(out.sum().item())}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
