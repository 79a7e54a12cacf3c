#!/usr/bin/env python3
"""Developer tool: time c2d_sat_rect_pairs_pose (and the vertex-format kernel beside it) on the config-2 workload with
HIP events on the kernel's stream, check a sample against the CPU oracle.  TEST INFRASTRUCTURE (uses oracle/).
usage: pose_bench.py [pairs] [reps];  C2D_LIBRARY=<other build of libc2d.so> for A/B runs."""
import os
import sys

import torch  # before libc2d.so
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from oracle import cpu as oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    eng = pkg.Engine(0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5A7)
    pose = torch.empty((10, n), dtype=torch.float32, device=dev)
    for r in range(2):
        pose[5 * r + 0].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 1].uniform_(-8.0, 8.0, generator=gen)
        pose[5 * r + 2].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 3].uniform_(0.1, 5.0, generator=gen)
        pose[5 * r + 4].uniform_(0.0, 2.0 * np.pi, generator=gen)
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    torch.cuda.synchronize()
    ptrs = [pose.data_ptr() + k * pose.stride(0) * 4 for k in range(10)]

    def step():
        eng.sat_rect_pairs_pose(ptrs, n, out.data_ptr(), cnt.data_ptr(), stream=sh)

    for _ in range(300):
        step()
    torch.cuda.synchronize()
    cnt.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{os.path.basename(pkg.library_path())}: {n} pairs  {ms * 1e3:.2f} us  {n / ms / 1e6:.2f} Gpairs/s  "
          f"{41 * n / ms / 1e6:.0f} GB/s ({41 * n / ms / 1e6 / 8000:.3f} of 8 TB/s)  collide rate {cnt.item() / reps / n:.4f}")
    m = min(n, 500_003)
    ref, _ = oracle.sat_rect_pairs_pose(pose[:, n - m:].contiguous().cpu().numpy())
    bad = int((ref != out[n - m:].cpu().numpy()).sum())
    print(f"oracle check on the last {m} pairs: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
