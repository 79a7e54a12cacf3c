"""Maximum-size checks, run as a separate process by tests/test_gpu_large.py (torch is used to
build multi-GB device inputs and has to be imported before libc2d.so).

  1. 1.2e9 rectangle pairs: every plane is 4.8 GB, so element offsets pass 2^32 bytes and the
     launch has 4.7e6 blocks.  The input is a 2e6-pair block repeated 600 times; the output must
     be 600 copies of the oracle's booleans for that block and the count 600 x the block's count.
  2. Monte-Carlo sample indices beyond 2^32: hits over [2^32 - 1e6, 2^32 + 1e6) must equal the
     oracle's, and a 6e9-sample range must equal the sum of its two halves.
TEST INFRASTRUCTURE: uses the oracle as the checker."""
import os
import sys

import torch
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
import importlib  # noqa: E402

wl = importlib.import_module("c2d_amd.workloads")
from oracle import cpu as oracle  # noqa: E402


def main():
    eng = pkg.Engine(0)
    dev = torch.device("cuda", 0)
    B, R = 2_000_000, 600
    n = B * R
    poses = wl.random_obb_pose_planes(B, seed=77, extent=4.0)
    block = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    ref, ref_cnt = oracle.sat_rect_pairs_verts(block)
    blk = torch.from_numpy(block).to(dev)
    planes = torch.empty((16, n), dtype=torch.float32, device=dev)
    planes.view(16, R, B)[:] = blk[:, None, :]
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    row = lambda t, k: t.data_ptr() + k * t.stride(0) * t.element_size()  # noqa: E731
    eng.sat_rect_pairs_verts([row(planes, k) for k in range(16)], n, out.data_ptr(), cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(cnt.item()) == R * ref_cnt, (int(cnt.item()), R * ref_cnt)
    refd = torch.from_numpy(ref).to(dev)
    assert bool((out.view(R, B) == refd[None, :]).all()), "booleans differ somewhere in the 1.2e9-pair batch"
    print(f"large SAT ok: {n} pairs, plane size {4 * n / 1e9:.1f} GB, count {int(cnt.item())}")
    del planes, out

    sc = wl.MC_PAIR_SCENE
    args = (sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 99, 7)
    hits = torch.zeros(1, dtype=torch.int64, device=dev)

    def gpu(begin, count):
        hits.zero_()
        eng.mc_pair(*args, begin, count, hits.data_ptr())
        torch.cuda.synchronize()
        return int(hits.item())

    b0 = (1 << 32) - 1_000_000
    assert gpu(b0, 2_000_000) == oracle.mc_pair(*args, b0, 2_000_000)
    whole = gpu(0, 6_000_000_000)
    assert whole == gpu(0, 3_000_000_001) + gpu(3_000_000_001, 2_999_999_999)
    print(f"large MC ok: 6e9 samples, p = {whole / 6e9:.6f}")


if __name__ == "__main__":
    main()
    print("large size ok")
