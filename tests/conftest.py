import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()
    _stamp_commit()


def _stamp_commit():
    """build/commit_stamp.txt = the commit under test, for the exploring fuzz legs on a box whose snapshot has no .git"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_seed", os.path.join(ROOT, "tests", "tools", "fuzz_seed.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.write_stamp()


# `pytest -x -m gpu` stops at the first failure: the oracle-parity tests of the hot path run FIRST, the
# tests whose outcome also depends on timing — subprocess launches, watchdog deadlines, a peer that never arrives — LAST, so that
# a flake there can never leave a parity row untested.  Files not named keep their place between the two groups.
GPU_ORDER_FIRST = ["test_gpu_sat.py", "test_gpu_sat_quads.py", "test_gpu_mc.py", "test_gpu_fullsize.py", "test_gpu_poly_binned.py", "test_gpu_large.py",
                   "test_gpu_mc_poly.py", "test_gpu_guard_bands.py", "test_gpu_host_batches.py", "test_gpu_fmad.py", "test_gpu_graph.py",
                   "test_gpu_fuzz_explore.py"]
GPU_ORDER_LAST = ["test_drivers.py", "test_gpu_workspace_guard.py", "test_gpu_dist.py"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        if item.get_closest_marker("gpu") is None:
            return (0, 0)                                     # CPU tests: untouched, in front (deselected under -m gpu)
        if name in GPU_ORDER_FIRST:
            return (1, GPU_ORDER_FIRST.index(name))
        if name in GPU_ORDER_LAST:
            return (3, GPU_ORDER_LAST.index(name))
        return (2, 0)

    items.sort(key=key)                                       # (stable: the order inside a file stays)


def _ensure_built():
    """The suite needs lib/libc2d.so, the CLI drivers and the oracle: build whatever is missing
    (a fresh checkout has none of them; hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess

    pkg_dir = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd")
    need = [os.path.join(pkg_dir, "lib", n) for n in ("libc2d.so", "libc2d_fmad1.so", "libc2d_fmad2.so", "libc2d_nopretest.so", "libc2d_movecheck.so")]
    need += [os.path.join(pkg_dir, "lib-rehearsal", "libc2d.so")]
    need += [os.path.join(ROOT, "oracle", n) for n in ("libc2d_oracle.so", "libc2d_oracle_fmad1.so", "libc2d_oracle_fmad2.so")]
    need += [os.path.join(pkg_dir, "bin", b) for b in ("generate_dataset", "compute_collision_probability", "ztest")]
    if all(os.path.exists(p) for p in need):
        return
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run(["make", "-C", ROOT, "-j4", "all"], check=True, env=env, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def wl(pkg):
    import importlib

    return importlib.import_module("c2d_amd.workloads")


@pytest.fixture(scope="session")
def eng(pkg):
    """One Engine (c2d_ctx) on device 0 for the whole GPU session.  No fallback:
    if the HIP library or the device is missing this raises."""
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu

    cpu.lib()
    # the box's CPU share, not the host's thread count: a 1-GPU job gets 16 cores of a host with many more hardware threads, and
    # an OpenMP team of the host's size on that quota turns every small oracle call into milliseconds of thread wake-ups
    cpu.set_num_threads(cpu.usable_cores())
    return cpu
