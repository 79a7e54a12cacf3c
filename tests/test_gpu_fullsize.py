"""BASELINE.json configurations 4 and 5 at the sizes bench.py runs them ("benchmarked, not tested").

  config 4, one GPU's shard: 4e6 data points, adaptive stopping with max_samples 120 000 —
      sharding invariance (the shard in one call == its two halves with scene_id_base set), total == sum(n_used),
      exact per-scene hit / sample counts against the oracle on 200 random blocks of 100 scenes spread over the shard
      (20 000 data points, about 2e9 samples: 15 s on the box's 16 cores), and the whole shard
      against the build that evaluates every sample in full (lib/libc2d_nopretest.so): the pretests and the
      compaction queue change no count at this size either;
  config 5: 1e7 polygon pairs — runs in tests/fullsize_poly_check.py (inputs are built with torch on the device).
  pretest boundary sweep (was csrc/tools/validate_pretest.py, reduced): scenes spread around the certain-miss
      boundary, shipped build == full-evaluation build."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NOPRETEST = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "lib", "libc2d_nopretest.so")


def _scenes_run(e, pkg, d_p, d_s, d_sc_ptr, ns, base, max_samples=120_000):
    d_h, d_u = e.zeros(ns, np.uint32), e.zeros(ns, np.uint32)
    total, iters = e.mc_scenes(d_p, 65536, d_s, 65536, d_sc_ptr, ns, 4.07, 1.74, (0.0, 0.01, 0.1, 1.0), (1e-4, 1e-3, 1e-2), max_samples, 11,
                               base, d_h, d_u, None)
    h, u = d_h.get(), d_u.get()
    d_h.free()
    d_u.free()
    return h, u, total, iters


def test_config4_shard_at_full_size(eng, pkg, oracle, wl):
    ns = 4_000_000
    seed = int(os.environ.get("C2D_FULLSIZE_SEED", "7"), 0)   # tables and scenes of another seed: profiles/r06_fullsize_seeds.sh
    tp, ts, _ = wl.random_tables(65536, 65536, seed=seed)
    d_p, d_s = eng.to_device(tp), eng.to_device(ts)
    d_sc = eng.empty(ns, pkg.SCENE_DT)
    eng.sample_scenes(d_p, 65536, d_s, 65536, 4.07, 1.74, 4.0, seed, 0, ns, d_sc)
    h, u, total, iters = _scenes_run(eng, pkg, d_p, d_s, d_sc, ns, 0)
    assert total == int(u.astype(np.int64).sum())
    assert iters >= 20 and int(u.max()) >= 120_000 and int(u.min()) == 1000      # early stoppers and scenes run to the cap
    # the same shard as two half-size calls (what two GPUs would do)
    half = ns // 2
    h0, u0, t0, _ = _scenes_run(eng, pkg, d_p, d_s, d_sc.ptr, half, 0)
    h1, u1, t1, _ = _scenes_run(eng, pkg, d_p, d_s, d_sc.ptr + half * pkg.SCENE_DT.itemsize, ns - half, half)
    assert np.array_equal(np.concatenate([h0, h1]), h) and np.array_equal(np.concatenate([u0, u1]), u) and t0 + t1 == total
    # exact oracle comparison on random blocks
    scenes = d_sc.get()
    rng = np.random.default_rng(4)
    checked = 0
    for b in rng.integers(0, ns - 100, 200):
        b = int(b)
        rh, ru, _, _ = oracle.mc_scenes(tp, ts, scenes[b:b + 100], 4.07, 1.74, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 120_000, 11, b)
        assert np.array_equal(rh, h[b:b + 100]) and np.array_equal(ru, u[b:b + 100]), b
        checked += int(ru.astype(np.int64).sum())
    assert checked > 1_000_000_000
    # the full-evaluation build on the whole shard
    full = pkg.Engine(0, lib_path=NOPRETEST)
    fh, fu, ft, _ = _scenes_run(full, pkg, d_p.ptr, d_s.ptr, d_sc.ptr, ns, 0)
    assert np.array_equal(fh, h) and np.array_equal(fu, u) and ft == total
    full.close()
    for a in (d_p, d_s, d_sc):
        a.free()


def test_pretest_boundary_sweep_against_full_evaluation(eng, pkg):
    full = pkg.Engine(0, lib_path=NOPRETEST)
    rng = np.random.default_rng(2025)
    N = 200_000_000
    d = eng.zeros(1, np.uint64)
    zero, some = 0, 0
    for i in range(300):
        w, h = rng.uniform(0.1, 5, 2)
        th = rng.uniform(0, 6.283)
        sd = tuple(np.sqrt(rng.uniform(0, 0.3, 3)).tolist()) + ((float(np.sqrt(rng.uniform(0, 0.3))), float(np.sqrt(rng.uniform(0, 0.3)))) if i % 3 == 0 else (0.0, 0.0))
        rho = np.hypot(w / 2 + 3.385 * sd[3], h / 2 + 3.385 * sd[4])
        dist = rho + rng.choice([0.87, 2.035]) + rng.uniform(-1.0, 4.0) * max(sd[0], sd[1], 0.05)
        ang = rng.uniform(0, 6.283)
        pos = (float(dist * np.cos(ang)), float(dist * np.sin(ang)))
        got = []
        for e in (eng, full):
            e.memset(d, 0, 8)
            e.mc_pair(4.07, 1.74, pos, (float(w), float(h), float(th)), sd, 777, i, 0, N, d)
            got.append(int(d.get()[0]))
        assert got[0] == got[1], (i, got)
        zero += got[0] == 0
        some += 0 < got[0] < N // 100
    assert zero > 20 and some > 50     # the sweep really covers the rare-collision regime
    d.free()
    full.close()


def test_config5_polygons_at_full_size():
    out = subprocess.run([sys.executable, os.path.join(HERE, "fullsize_poly_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "fullsize poly ok" in out.stdout
