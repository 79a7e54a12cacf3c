import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


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
