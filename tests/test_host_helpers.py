"""CPU test of the C++ host helpers (npy I/O, flag parser, driver_common) — compiles and runs
tests/cpp/test_host_helpers.cpp, then cross-checks the .npy files it wrote with numpy."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_helpers(tmp_path):
    exe = tmp_path / "test_host_helpers"
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "cpp", "test_host_helpers.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    a = np.load(tmp_path / "a1000.npy")
    assert a.shape == (1000, 5) and a.dtype == np.float32 and a[3, 2] == np.float32(0.25 * 17 - 3.0)
    assert np.load(tmp_path / "a0.npy").shape == (0, 5)
    assert np.load(tmp_path / "one.npy").tolist() == [1.5, -2.5, 3.5]
    # numpy's own histogram agrees with the edges used by the drivers' summary
    cp = np.array([0.0, 0.0005, 0.001, 0.0099, 0.01, 0.05, 0.1, 0.5, 1.0], np.float32)
    assert np.histogram(cp, [0, 0.001, 0.01, 0.1, 1])[0].tolist() == [2, 3, 1, 3]


def test_bench_self_launch_is_refused_under_a_preloaded_profiler():
    """`bench.py --gpus N` without a launcher starts its ranks by fork + exec: refused when a profiler / tool library is preloaded
    into the parent (it has initialised the GPU before main; an exec from such a process is what the pool's hosts forbid) — the
    same rule as the drivers' --gpus N (tests/test_drivers.py).  Nothing is started: no torch import, no rendezvous."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["ROCP_TOOL_LIBRARIES"] = "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 2 and "refused" in out.stderr and "ROCP_TOOL_LIBRARIES" in out.stderr and not out.stdout.strip(), out.stderr
