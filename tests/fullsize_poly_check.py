"""BASELINE config 5 at full size, run as a separate process by tests/test_gpu_fullsize.py (the 1e7-pair input is
built with torch on the device; torch has to be imported before libc2d.so).

Size-independent properties of the polygon SAT on 1e7 pairs (K ~ U{3..16}, the bench.py workload):
  * count == sum(out);
  * swapping polygons A and B changes no boolean;
  * rotating the vertex list of every polygon (cyclic shift by one) changes no boolean — same edge set, same
    projections, in another order;
  * reversing the orientation (clockwise lists) changes no boolean;
  * exact agreement with the oracle on ALL 1e7 pairs.
TEST INFRASTRUCTURE: uses the oracle as the checker."""
import os
import sys

import torch
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import torch_random_convex_polygons, KMAX  # noqa: E402

pkg = load_package()
from oracle import cpu as oracle  # noqa: E402


def main():
    eng = pkg.Engine(0)
    dev = torch.device("cuda", 0)
    n = 10_000_000
    seed = int(os.environ.get("C2D_FULLSIZE_SEED", "0xC0FFEE"), 0)    # (another seed: profiles/r06_fullsize_seeds.sh)
    vx, vy, kk = torch_random_convex_polygons(torch, dev, n, seed=seed)
    torch.cuda.synchronize()

    def run(ax, ay, ak):
        out = torch.zeros(n, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.sat_poly_pairs(ax.data_ptr(), ay.data_ptr(), ak.data_ptr(), n, out.data_ptr(), cnt.data_ptr())
        eng.synchronize()
        return out, int(cnt.item())

    out, cnt = run(vx, vy, kk)
    assert cnt == int(out.sum(dtype=torch.int64).item())
    assert 0.05 < cnt / n < 0.07
    # A <-> B
    o2, c2 = run(vx.flip(0).contiguous(), vy.flip(0).contiguous(), kk.flip(0).contiguous())
    assert c2 == cnt and bool((o2 == out).all())
    # cyclic shift by one inside each polygon's k vertices; reversal of the orientation
    idx = torch.arange(KMAX, device=dev)[None, :, None]                 # [1, KMAX, 1]
    k3 = kk.to(torch.int64)[:, None, :]                                 # [2, 1, n]
    shift = torch.where(idx < k3, (idx + 1) % k3, idx)
    o3, c3 = run(torch.gather(vx, 1, shift), torch.gather(vy, 1, shift), kk)
    assert c3 == cnt and bool((o3 == out).all())
    rev = torch.where(idx < k3, k3 - 1 - idx, idx)
    o4, c4 = run(torch.gather(vx, 1, rev), torch.gather(vy, 1, rev), kk)
    assert c4 == cnt and bool((o4 == out).all())
    # the oracle on the WHOLE batch ("boolean equality on the full 10^7 set"), ~2 s of OpenMP
    del o2, o3, o4, shift, rev
    oracle.set_num_threads(oracle.usable_cores())
    ref, ref_cnt = oracle.sat_poly_pairs(vx.cpu().numpy(), vy.cpu().numpy(), kk.cpu().numpy())
    got = out.cpu().numpy()
    assert ref.shape == got.shape == (n,)
    assert np.array_equal(got, ref) and ref_cnt == cnt
    eng.check_async()
    print(f"fullsize poly ok (seed {seed}): {n} pairs, {cnt} colliding, booleans equal to the oracle's on {int((got == ref).sum())} of {n}")


if __name__ == "__main__":
    main()
