"""world_size-2, -3 and -8 gloo tests of the multi-GPU path's host logic: range
sharding + one all-reduce of the counters.  The per-rank compute here is the CPU oracle
(test stand-in for the HIP kernels, which need a GPU); the property under test is that
shards + reduce reproduce the single-process result exactly."""
import os
import socket

import numpy as np
import pytest

# torch is imported inside the tests: the GPU run (-m gpu) collects this module too and should
# not pay for (or be affected by) torch's bundled HIP runtime.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys

    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package

    load_package()
    import importlib

    sh = importlib.import_module("c2d_amd.sharding")
    wl = importlib.import_module("c2d_amd.workloads")
    from oracle import cpu as oracle

    # (1) MC single pair: sample index space split over ranks, one all-reduce of the hits
    S = 200_001
    b, e = sh.shard_range(S, rank, world)
    sc = wl.MC_PAIR_SCENE
    local = oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, b, e - b)
    hits, samples = sh.all_reduce_counters([local, e - b])

    # (2) pair batch: contiguous range partition, count reduced
    n = 30_001
    poses = wl.random_obb_pose_planes(n, seed=3, extent=3.0)
    pb, pe = sh.shard_range(n, rank, world)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5, pb:pe]), oracle.rects_from_poses(*poses[5:, pb:pe])])
    out, cnt = oracle.sat_rect_pairs_verts(planes)
    (total_cnt,) = sh.all_reduce_counters([cnt])

    # (3) scenes: scene ranges per rank with scene_id_base = range begin
    tp, ts, _ = wl.random_tables(16, 16, seed=4)
    scenes = oracle.sample_scenes(tp, ts, 4.07, 1.74, 4.0, 5, 0, 41)
    sb, se = sh.shard_range(41, rank, world)
    h, u, _, tot = oracle.mc_scenes(tp, ts, scenes[sb:se], 4.07, 1.74, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 2000, 8, sb)
    scene_hits, scene_samples = sh.all_reduce_counters([int(h.sum()), tot])
    tmax = sh.max_over_ranks(1.0 + rank)
    # (4) what bench.py adds to an N > 1 line: every rank's kernel time in rank order, the rate of the kernels alone and
    # the slowest rank's roofline fraction.  Rank r "measures" 0.1 + 0.01 r ms on 1000 + r units of which 10 (r + 1) count as work.
    rows = sh.gather_rows([rank, 10.0 * rank])
    spread, kernels_only, frac_slowest = sh.kernel_time_spread(0.1 + 0.01 * rank, 1000 + rank, frac=0.5, work=10 * (rank + 1))
    if rank == 0:
        q.put((hits, samples, total_cnt, scene_hits, scene_samples, tmax, rows, spread, kernels_only, frac_slowest))
    dist.destroy_process_group()


def test_shard_range_partitions_exactly(pkg):
    import importlib

    sh = importlib.import_module("c2d_amd.sharding")
    for total in (0, 1, 7, 8, 10**7, 10**8 + 3):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.shard_range(10, 2, 2)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_sharding_reproduces_single_process(oracle, wl, world):
    """2, 3 and 8 ranks (the arithmetic of shard_range, scene_id_base and the counter reduce at the world size of the first real
    8-GPU lease, with a world that does not divide the work and one with fewer scenes per rank than ranks)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hits, samples, total_cnt, scene_hits, scene_samples, tmax, rows, spread, kernels_only, frac_slowest = res
    assert rows == [[float(r), 10.0 * r] for r in range(world)]                       # rank order
    slowest = world - 1
    assert spread["ranks"] == world and spread["slowest_rank"] == slowest
    assert spread["min"] == 0.1 and spread["max"] == round(0.1 + 0.01 * slowest, 5) and spread["min"] <= spread["median"] <= spread["max"]
    assert abs(kernels_only - sum(1000 + r for r in range(world)) / ((0.1 + 0.01 * slowest) * 1e-3)) < 1e-3
    # rank 0's fraction 0.5 (10 work units in 0.1 ms) rescaled to the slowest rank's work and time
    assert frac_slowest == round(0.5 * (10 * world / 10) * (0.1 / (0.1 + 0.01 * slowest)), 4)
    sc = wl.MC_PAIR_SCENE
    assert samples == 200_001
    assert hits == oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, 200_001)
    poses = wl.random_obb_pose_planes(30_001, seed=3, extent=3.0)
    planes = np.concatenate([oracle.rects_from_poses(*poses[:5]), oracle.rects_from_poses(*poses[5:])])
    assert total_cnt == oracle.sat_rect_pairs_verts(planes)[1]
    tp, ts, _ = wl.random_tables(16, 16, seed=4)
    scenes = oracle.sample_scenes(tp, ts, 4.07, 1.74, 4.0, 5, 0, 41)
    h, u, _, tot = oracle.mc_scenes(tp, ts, scenes, 4.07, 1.74, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 2000, 8, 0)
    assert scene_hits == int(h.sum()) and scene_samples == tot
    assert tmax == float(world)
