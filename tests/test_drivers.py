"""Tests of the C++ host drivers (the reference's CLI surface).
CPU part: flag parsing, .npy files readable by numpy, loud failure without a GPU.
End-to-end runs whose rows equal the CPU oracle's come in two flavours of the same test: `[gpu]` = the shipped binaries on
libc2d.so (`-m gpu`), `[double]` = the same driver sources compiled against tests/cpp/c2d_cpu_double.cpp, a TEST DOUBLE of the C-ABI
that hands the Monte-Carlo work to the oracle — so that the drivers' host logic (tables and their files, batch numbering, scene-id
bases, the deal of batches over ranks, shuffles, summaries) is exercised in the CPU suite too.  The double says nothing about the
kernels; it is never built into, linked with or selectable by the product."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "bin")
GEN = os.path.join(BIN, "generate_dataset")
CCP = os.path.join(BIN, "compute_collision_probability")
ZT = os.path.join(BIN, "ztest")


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)


@pytest.fixture(scope="session")
def double_dir(tmp_path_factory):
    """generate_dataset, compute_collision_probability and ztest compiled against the CPU test double"""
    d = tmp_path_factory.mktemp("drivers_on_the_cpu_double")
    host = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc", "host")
    oracle_dir = os.path.join(ROOT, "oracle")
    procs = [subprocess.Popen(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), os.path.join(host, name + ".cpp"),
                               os.path.join(ROOT, "tests", "cpp", "c2d_cpu_double.cpp"), "-o", str(d / name), "-L" + oracle_dir, "-lc2d_oracle",
                               "-Wl,-rpath," + oracle_dir]) for name in ("generate_dataset", "compute_collision_probability", "ztest")]
    assert all(p.wait() == 0 for p in procs)
    return d


@pytest.fixture(params=[pytest.param("gpu", marks=pytest.mark.gpu), "double"])
def drv(request):
    """the three drivers: the shipped binaries (needs the GPU) or the ones on the CPU double"""
    import types

    if request.param == "gpu":
        return types.SimpleNamespace(kind="gpu", GEN=GEN, CCP=CCP, ZT=ZT, run=run)
    d = request.getfixturevalue("double_dir")
    request.getfixturevalue("oracle")   # (builds the oracle library if missing)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    return types.SimpleNamespace(kind="double", GEN=str(d / "generate_dataset"), CCP=str(d / "compute_collision_probability"), ZT=str(d / "ztest"),
                                 run=lambda cmd, **kw: run(cmd, env=dict(env, **kw.pop("env", {})), **kw))


def test_help_lists_every_reference_flag():
    out = run([GEN, "--help"])
    assert out.returncode == 1  # the reference exits with 1 after --help (generate_dataset.cu:96-99)
    for flag in ["data_dir", "num_batches", "batch_size", "start_batch_count", "num_poses", "num_variances", "shape_variance",
                 "max_samples", "accuracy_bins", "bin_accuracy", "min_variance", "max_variance", "min_pose", "max_pose",
                 "robot_width", "robot_height", "spread", "pose_dir", "variance_dir"]:
        assert "--" + flag in out.stdout, flag
    for short in ["-n [", "-b [", "-s [", "-w [", "-h ["]:
        assert short in out.stdout
    out = run([CCP, "--help"])
    assert out.returncode == 1
    for flag in ["data_in", "data_out", "max_samples", "robot_width", "robot_height", "shuffle"]:
        assert "--" + flag in out.stdout, flag


def test_bad_flags_fail_cleanly(tmp_path):
    assert run([GEN, "--no_such_flag", "1"]).returncode != 0
    assert run([GEN, "--min_pose", "1", "2"]).returncode != 0          # needs 3 values
    assert run([GEN, "--data_dir", str(tmp_path), "--bin_accuracy", "0.1"]).returncode != 0
    assert run([CCP, "--data_in", str(tmp_path / "nope"), "--data_out", str(tmp_path / "nope2")]).returncode != 0


def test_multi_rank_flags_are_checked_before_any_gpu_work(tmp_path):
    """Values all ranks must agree on are resolved once: hand-launched ranks without an aggregation link
    must be given them; --gpus N (the driver launches its own ranks) excludes --rank / --world_size."""
    out = run([GEN, "--data_dir", str(tmp_path / "a"), "--rank", "1", "--world_size", "2", "-n", "2", "-b", "10"])
    assert out.returncode != 0 and "--seed" in out.stderr
    out = run([CCP, "--data_in", str(tmp_path), "--data_out", str(tmp_path), "--rank", "1", "--world_size", "2"])
    assert out.returncode != 0 and "--start_batch_count" in out.stderr
    for tool in (GEN, CCP):
        out = run([tool, "--gpus", "2", "--rank", "0", "--world_size", "2"])
        assert out.returncode != 0 and "starts the ranks itself" in out.stderr
        assert run([tool, "--gpus", "0"]).returncode != 0
    assert "--gpus" in run([GEN, "--help"]).stdout and "--pair_samples" in run([CCP, "--help"]).stdout


def test_self_launch_is_refused_under_a_preloaded_profiler(tmp_path):
    """--gpus N starts the ranks by fork + execv of the driver itself: safe only from a process that has not initialised the GPU.
    A profiler / tool library preloaded into the launcher (rocprofv3 puts its own into every process it starts) has done that
    before main, and an exec from such a process is what the pool's hosts forbid.  The launcher refuses, names the variable and
    says what to do instead (profile one rank); nothing is started, no id file is left.  A single rank is not affected."""
    for var, val in (("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"), ("HSA_TOOLS_LIB", "libanything.so"),
                     ("LD_PRELOAD", "/nonexistent/libRocProfiler-sdk-tool.so.1")):
        env = dict(os.environ, TMPDIR=str(tmp_path))
        env[var] = val
        for cmd in ([GEN, "--data_dir", str(tmp_path / "d"), "--gpus", "2", "-n", "2", "-b", "10"], [CCP, "--pair_samples", "1000", "--gpus", "3"]):
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=60, env=env)
            assert out.returncode != 0 and "refused" in out.stderr and var in out.stderr and "--rank k --world_size" in out.stderr, out.stderr
            assert "no usable device" not in out.stderr and "execv" not in out.stderr.replace("by execv", "")   # no rank was started
        assert not list(tmp_path.glob("c2d_dist_id_*")) and not (tmp_path / "d").exists()
    # an unrelated preload does not trigger it (the ranks start and, on this box, fail for want of a GPU or run)
    assert "--gpus" in run([GEN, "--help"]).stdout and "never the launcher" in run([GEN, "--help"]).stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/rocprofv3"), reason="needs rocprofv3")
def test_self_launch_is_refused_under_the_real_rocprofv3(tmp_path):
    """The same with the profiler itself in front, not a hand-set variable: what rocprofv3 puts into its child's environment is what
    the launchers look for — the driver's `--gpus N` and `bench.py --gpus N` both refuse, and rocprofv3 hands their status on."""
    import sys

    env = dict(os.environ, TMPDIR=str(tmp_path))
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    prof = ["/opt/rocm/bin/rocprofv3", "--kernel-trace", "-d", str(tmp_path / "prof"), "--"]
    out = subprocess.run(prof + [GEN, "--data_dir", str(tmp_path / "d"), "--gpus", "2", "-n", "2", "-b", "10"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=str(tmp_path))
    assert out.returncode != 0 and "--gpus 2 refused" in out.stderr and "rocprofiler" in out.stderr, out.stderr[-2000:]
    assert not (tmp_path / "d").exists() and not list(tmp_path.glob("c2d_dist_id_*"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(prof + [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         env=env, cwd=str(tmp_path))
    assert out.returncode != 0 and "--gpus 2 refused" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stderr[-2000:]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-only behaviour")
def test_self_launched_ranks_propagate_failure_without_gpu(tmp_path):
    """--gpus N: the launcher starts N fresh copies of itself and returns the first non-zero exit status; without a GPU every
    rank fails loudly at c2d_ctx_create, and so does the launcher."""
    out = run([GEN, "--data_dir", str(tmp_path / "d"), "--gpus", "2", "-n", "2", "-b", "10", "--num_poses", "10", "--num_variances", "10"])
    assert out.returncode != 0 and out.stderr.count("no usable device") >= 1   # the launcher stops the other ranks as soon as one fails
    out = run([CCP, "--pair_samples", "1000", "--gpus", "3"])
    assert out.returncode != 0 and out.stderr.count("no usable device") >= 1
    assert not list(tmp_path.glob("c2d_dist_id_*"))


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-only behaviour")
def test_generate_dataset_fails_loudly_without_gpu(tmp_path):
    """no CPU fallback anywhere: since round 4 even the tables are drawn on the device (c2d_uniform_table_minstd), so without a GPU
    the driver stops at c2d_ctx_create and writes no table"""
    d = tmp_path / "data"
    out = run([GEN, "--data_dir", str(d), "-n", "1", "-b", "100", "--num_poses", "1000", "--num_variances", "500"])
    assert out.returncode != 0 and "no usable device" in out.stderr
    assert not (d / "poses.npy").exists() and not (d / "variances.npy").exists()


def test_generate_dataset_tables_and_meta_files(tmp_path, drv):
    d = tmp_path / "data"
    out = drv.run([drv.GEN, "--data_dir", str(d), "-n", "1", "-b", "100", "--num_poses", "1000", "--num_variances", "500", "--max_samples", "2000",
               "--min_pose", "0.5", "0.25", "0", "--max_pose", "2", "3", "1", "--accuracy_bins", "0", "0.5", "1",
               "--bin_accuracy", "0.01", "0.02", "--seed", "1"])
    assert out.returncode == 0, out.stderr
    poses = np.load(d / "poses.npy")
    var = np.load(d / "variances.npy")
    assert poses.shape == (1000, 3) and poses.dtype == np.float32
    assert var.shape == (500, 5) and var.dtype == np.float32
    assert (poses[:, 0] >= 0.5).all() and (poses[:, 0] <= 2).all() and (poses[:, 1] <= 3).all() and (poses[:, 2] <= 1).all()
    assert (var[:, :3] >= 0).all() and (var[:, :3] <= 0.3 + 1e-7).all()
    assert (var[:, 3:] == 0).all()                                      # shape_variance off (generate_dataset.cu:285-290)
    assert np.load(d / "meta" / "accuracy_bins.npy").tolist() == [0, 0.5, 1]
    assert np.allclose(np.load(d / "meta" / "bin_accuracy.npy"), [0.01, 0.02])
    # the first values of libstdc++'s default engine through uniform_real_distribution<float>(0, 0.3): 16807, 16807^2 mod M, ...
    M = 2147483647
    x, want = 1, []
    for _ in range(3):
        x = x * 16807 % M
        want.append(np.float32(np.float32(x - 1) * np.float32(2.0 ** -31)) * np.float32(0.3))
    assert var[0, :3].tolist() == [float(w) for w in want]
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "tables_on_device" in summary["phases_s"] and "tables_save_npy_thread" in summary["phases_s"]


@pytest.mark.gpu
def test_device_tables_equal_the_reference_loop(tmp_path):
    """c2d_uniform_table_minstd / c2d_sqrt_f32 against std::default_random_engine + uniform_real_distribution<float> (the reference's own
    loop, generate_dataset.cu:279-332), bit for bit: small and odd sizes, and the reference's default 64^4 rows per table"""
    exe = tmp_path / "test_device_tables"
    lib = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "test_device_tables.cpp"), "-o", str(exe),
                    "-L" + lib, "-lc2d", "-Wl,-rpath," + lib, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    for nv, np_, shape in [(1, 1, 0), (77, 90, 1), (70001, 5003, 0), (64 ** 4, 64 ** 4, 0)]:
        out = subprocess.run([str(exe), str(nv), str(np_), str(shape)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and " 0 differing floats" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-only behaviour")
def test_ccp_reads_numpy_written_inputs_then_fails_loudly_without_gpu(tmp_path):
    din, dout = tmp_path / "in", tmp_path / "out"
    (dout / "meta").mkdir(parents=True)
    din.mkdir()
    rng = np.random.default_rng(0)
    np.save(dout / "poses.npy", rng.uniform(0.1, 5, (10, 3)).astype(np.float32))
    np.save(dout / "variances.npy", rng.uniform(0, 0.3, (10, 5)).astype(np.float32))
    np.save(dout / "meta" / "accuracy_bins.npy", np.array([0, .01, .1, 1], np.float32))
    np.save(dout / "meta" / "bin_accuracy.npy", np.array([1e-4, 1e-3, 1e-2], np.float32))
    np.save(din / "0.npy", rng.uniform(0, 5, (50, 4)).astype(np.float32))
    out = run([CCP, "--data_in", str(din), "--data_out", str(dout)])
    assert "num poses: 10" in out.stdout and "num data points: 50" in out.stdout
    assert out.returncode != 0 and "no usable device" in out.stderr


# ---- GPU end-to-end ---------------------------------------------------------------------------

def test_generate_dataset_end_to_end_matches_oracle(tmp_path, oracle, drv):
    d = tmp_path / "data"
    B, NB = 1500, 2
    out = drv.run([drv.GEN, "--data_dir", str(d), "-n", str(NB), "-b", str(B), "-s", "3", "--num_poses", "200", "--num_variances", "100",
               "--max_samples", "3000", "--seed", "77", "--shape_variance", "--spread", "3.5"])
    assert out.returncode == 0, out.stderr
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    poses = np.load(d / "poses.npy")
    var = np.load(d / "variances.npy")
    sd = np.sqrt(var).astype(np.float32)
    assert sorted(p.name for p in d.glob("[0-9]*.npy")) == ["3.npy", "4.npy"]   # --start_batch_count numbering
    total = 0
    for b in range(NB):
        rows = np.load(d / f"{3 + b}.npy")
        assert rows.shape == (B, 5) and rows.dtype == np.float32
        base = (3 + b) * B
        scenes = oracle.sample_scenes(poses.view(oracle.POSE_DT).reshape(-1), sd.view(oracle.STD_DT).reshape(-1), 4.07, 1.74, 3.5, 77, base, B)
        hits, used, ref_rows, tot = oracle.mc_scenes(poses.view(oracle.POSE_DT).reshape(-1), sd.view(oracle.STD_DT).reshape(-1), scenes,
                                                     4.07, 1.74, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 3000, 77, base)
        total += tot
        ref = ref_rows.view(np.float32).reshape(B, 5)
        # the file is the oracle's rows in shuffled order: compare as sorted multisets, bit for bit
        key = lambda a: a[np.lexsort(a.T[::-1])].view(np.uint32)  # noqa: E731
        assert np.array_equal(key(rows), key(ref))
        assert not np.array_equal(rows, ref)                        # it was shuffled (generate_dataset.cu:496)
    assert summary["mc_samples"] == total and summary["scenes"] == B * NB
    allrows = np.concatenate([np.load(d / f"{3 + b}.npy") for b in range(NB)])
    assert summary["cp_hist"] == np.histogram(allrows[:, 2], [0, 0.001, 0.01, 0.1, 1])[0].tolist()   # balance_datasets.py:49


def test_ccp_end_to_end_matches_oracle_and_continues_numbering(tmp_path, oracle, wl, drv):
    din, dout = tmp_path / "in", tmp_path / "out"
    (dout / "meta").mkdir(parents=True)
    din.mkdir()
    poses, sds, var = wl.random_tables(50, 40, seed=3)
    np.save(dout / "poses.npy", poses.view(np.float32).reshape(-1, 3))
    np.save(dout / "variances.npy", var)
    np.save(dout / "meta" / "accuracy_bins.npy", np.array([0, .01, .1, 1], np.float32))
    np.save(dout / "meta" / "bin_accuracy.npy", np.array([1e-4, 1e-3, 1e-2], np.float32))
    np.save(dout / "0.npy", np.zeros((3, 5), np.float32))           # an existing batch: numbering continues at 1
    N = 700
    rng = np.random.default_rng(1)
    batches = []
    for k in range(2):
        s = np.empty((N, 4), np.float32)
        s[:, 0] = rng.uniform(-6, 6, N)
        s[:, 1] = rng.uniform(-6, 6, N)
        s[:, 2] = rng.integers(0, 40, N)
        s[:, 3] = rng.integers(0, 50, N)
        np.save(din / f"{k}.npy", s)
        batches.append(s)
    out = drv.run([drv.CCP, "--data_in", str(din), "--data_out", str(dout), "--max_samples", "4000", "--shuffle", "false", "--seed", "9"])
    assert out.returncode == 0, out.stderr
    sd_from_var = np.sqrt(var).astype(np.float32).view(oracle.STD_DT).reshape(-1)
    for k in range(2):
        rows = np.load(dout / f"{1 + k}.npy")
        assert rows.shape == (N, 5)
        scenes = batches[k].view(oracle.SCENE_DT).reshape(-1)
        _, _, ref_rows, _ = oracle.mc_scenes(poses, sd_from_var, scenes, 4.07, 1.74, [0, .01, .1, 1], [1e-4, 1e-3, 1e-2], 4000, 9, (1 + k) * N)
        assert np.array_equal(rows.view(np.uint32), ref_rows.view(np.uint32).reshape(N, 5))   # input order kept (--shuffle false)


def test_ztest_single_file_modes(tmp_path, oracle, wl, drv):
    """ztest: explicit files, default meta written, constant 10000-sample schedule, --cps_only."""
    d = tmp_path / "data"
    d.mkdir()
    poses, sds, var = wl.random_tables(30, 20, seed=5)
    np.save(d / "poses.npy", poses.view(np.float32).reshape(-1, 3))
    np.save(d / "variances.npy", var)
    N = 400
    rng = np.random.default_rng(2)
    s = np.empty((N, 4), np.float32)
    s[:, 0] = rng.uniform(-6, 6, N)
    s[:, 1] = rng.uniform(-6, 6, N)
    s[:, 2] = rng.integers(0, 20, N)
    s[:, 3] = rng.integers(0, 30, N)
    np.save(tmp_path / "in.npy", s)
    sd_from_var = np.sqrt(var).astype(np.float32).view(oracle.STD_DT).reshape(-1)
    _, used, ref_rows, _ = oracle.mc_scenes(poses, sd_from_var, s.view(oracle.SCENE_DT).reshape(-1), 4.07, 1.74, [0, .01, .1, 1],
                                            [1e-4, 1e-3, 1e-2], 30000, 4, 0, schedule=(10000, 10000, 0))
    assert set(np.unique(used).tolist()) <= {10000, 20000, 30000} and len(np.unique(used)) > 1
    out = drv.run([drv.ZT, "--data_dir", str(d), "--data_file_in", str(tmp_path / "in.npy"), "--data_file_out", str(tmp_path / "rows.npy"),
               "--max_samples", "30000", "--shuffle", "false", "--seed", "4"])
    assert out.returncode == 0, out.stderr + out.stdout
    assert np.allclose(np.load(d / "meta" / "accuracy_bins.npy"), [0, 0.01, 0.1, 1])
    rows = np.load(tmp_path / "rows.npy")
    assert np.array_equal(rows.view(np.uint32), ref_rows.view(np.uint32).reshape(N, 5))
    out = drv.run([drv.ZT, "--data_dir", str(d), "--data_file_in", str(tmp_path / "in.npy"), "--data_file_out", str(tmp_path / "cps.npy"),
               "--max_samples", "30000", "--shuffle", "false", "--seed", "4", "--cps_only", "true", "--meta_dir", str(d / "meta")])
    assert out.returncode == 0, out.stderr + out.stdout
    cps = np.load(tmp_path / "cps.npy")
    assert cps.shape == (N,) and np.array_equal(cps.view(np.uint32), ref_rows["cp"].view(np.uint32))


def test_ranks_deal_batches_like_one_rank_on_the_double(tmp_path, wl, double_dir, oracle):
    """The N > 1 host logic of both drivers on the CPU double (the GPU flavour of this test, with the rehearsal build of the
    library, is tests/test_gpu_dist.py): `--gpus 2` and `--gpus 3` write the very batch files one rank writes, one summary
    aggregated over the ranks, the reference's numbering continued; config 3 split by sample index sums to the one-rank count."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    gen, ccp = str(double_dir / "generate_dataset"), str(double_dir / "compute_collision_probability")
    summary = lambda out: json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])  # noqa: E731
    common = ["-n", "5", "-b", "400", "-s", "2", "--num_poses", "60", "--num_variances", "40", "--max_samples", "3000", "--seed", "11"]
    outs = {}
    for w in (1, 2, 3):
        d = tmp_path / f"gen{w}"
        o = run([gen, "--data_dir", str(d)] + common + (["--gpus", str(w)] if w > 1 else []), env=env)
        assert o.returncode == 0, o.stderr + o.stdout
        outs[w] = (d, summary(o))
        assert [ln for ln in o.stdout.splitlines() if ln.startswith("{")].__len__() == 1          # one summary, rank 0's
    d1, s1 = outs[1]
    for w in (2, 3):
        d, s = outs[w]
        assert s["aggregated_over_ranks"] == w and s["world_size"] == w
        for key in ("batches", "scenes", "mc_samples", "hits", "cp_hist"):
            assert s[key] == s1[key], key
        assert sorted(p.name for p in d.glob("[0-9]*.npy")) == [f"{k}.npy" for k in range(2, 7)]
        for k in range(2, 7):
            assert np.array_equal(np.load(d1 / f"{k}.npy").view(np.uint32), np.load(d / f"{k}.npy").view(np.uint32)), (w, k)
        assert np.array_equal(np.load(d1 / "poses.npy"), np.load(d / "poses.npy"))
    # compute_collision_probability on generate_dataset's own scenes, with an existing batch in the output directory
    din = tmp_path / "in"
    din.mkdir()
    for k in range(2, 7):
        a = np.load(d1 / f"{k}.npy")
        np.save(din / f"{k - 2}.npy", np.ascontiguousarray(a[:, [0, 1, 3, 4]]))
    res = {}
    for w in (1, 3):
        dout = tmp_path / f"ccp{w}"
        (dout / "meta").mkdir(parents=True)
        for f in ("poses.npy", "variances.npy"):
            np.save(dout / f, np.load(d1 / f))
        for f in ("accuracy_bins.npy", "bin_accuracy.npy"):
            np.save(dout / "meta" / f, np.load(d1 / "meta" / f))
        np.save(dout / "0.npy", np.zeros((3, 5), np.float32))
        o = run([ccp, "--data_in", str(din), "--data_out", str(dout), "--max_samples", "3000", "--seed", "4"] + (["--gpus", str(w)] if w > 1 else []), env=env)
        assert o.returncode == 0, o.stderr + o.stdout
        res[w] = (dout, summary(o))
    for k in range(1, 6):
        assert np.array_equal(np.load(res[1][0] / f"{k}.npy").view(np.uint32), np.load(res[3][0] / f"{k}.npy").view(np.uint32)), k
    assert res[3][1]["aggregated_over_ranks"] == 3 and res[3][1]["mc_samples"] == res[1][1]["mc_samples"] and res[3][1]["hits"] == res[1][1]["hits"]
    sc = wl.MC_PAIR_SCENE
    S = 300_001
    ref = oracle.mc_pair(sc["robot_w"], sc["robot_h"], sc["pos"], sc["pose"], sc["std_dev"], 1234, 0, 0, S)
    for w in (1, 4):
        o = run([ccp, "--pair_samples", str(S), "--seed", "1234"] + (["--gpus", str(w)] if w > 1 else []), env=env)
        assert o.returncode == 0, o.stderr
        s = summary(o)
        assert s["hits"] == ref and s["samples"] == S and s["aggregated_over_ranks"] == w
