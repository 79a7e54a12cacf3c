"""The exploring legs of the differential fuzzers: the same generators as the fixed-seed legs (test_gpu_sat.py
test_poly_differential_fuzz, test_gpu_mc.py test_mc_differential_fuzz, test_gpu_poly_binned.py test_binned_differential_fuzz,
tests/tools/pose_fuzz.py; and tests/tools/verts_fuzz.py for the headline vertex-format entry points), seeded from the commit under test (tests/tools/fuzz_seed.py) and run for a time budget instead of
a configuration count, so that every run of the suite at a new commit meets inputs no earlier run has met.  Round 5's one
defect — the binning pass's move kernel reading past its arrays on a last partial tile — passed two green runs of the
fixed-seed suite and was met by a soak outside it (profiles/notes_r05_move_kernel_overread.md).

Every leg prints its seed BEFORE it starts (past pytest's capture: a GPU fault takes the captured output down with the
process) and names every configuration, before it runs, in gpurun_out/fuzz_trace/<leg>.txt.  To reproduce a failure:
    C2D_FUZZ_SEED=<base seed printed> python -m pytest tests/test_gpu_fuzz_explore.py -m gpu -k <leg>
($C2D_FUZZ_SECONDS, default 40, is each leg's budget; a configuration is a function of (seed, its index) alone.)"""
import importlib.util
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LEGS = ("sat_rect_pose", "sat_poly_rows", "sat_poly_binned", "mc_scenes", "sat_rect_verts")   # (append only: a leg's seed offset is its index)
TOOL = {"sat_rect_verts": "verts_fuzz", "sat_rect_pose": "pose_fuzz", "sat_poly_rows": "poly_fuzz", "sat_poly_binned": "binned_fuzz", "mc_scenes": "mc_fuzz"}


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("leg", LEGS)
def test_differential_fuzz_at_this_commits_seed(eng, oracle, capsys, leg):   # (`oracle`: sizes the OpenMP team to the box's CPU share)
    base, origin = _tool("fuzz_seed").commit_seed()
    seed = (base + 0x3C6EF35F * LEGS.index(leg)) & 0x7FFFFFFF   # one stream per leg
    budget = float(os.environ.get("C2D_FUZZ_SECONDS", "40"))
    fz = _tool(TOOL[leg])
    trace_dir = os.path.join(ROOT, "gpurun_out", "fuzz_trace")
    try:
        os.makedirs(trace_dir, exist_ok=True)
        trace = open(os.path.join(trace_dir, leg + ".txt"), "w")
    except OSError:   # a read-only checkout: the trace goes where the box lets it
        import tempfile

        trace_dir = tempfile.mkdtemp(prefix="c2d_fuzz_trace_")
        trace = open(os.path.join(trace_dir, leg + ".txt"), "w")
    with capsys.disabled():
        print(f"\n[fuzz] {leg}: seed {seed} = base seed {base} ({origin}) + leg offset, {budget:.0f} s; "
              f"reproduce with C2D_FUZZ_SEED={base}; configurations named in {trace.name}", flush=True)
    trace.write(f"# {leg}: seed {seed}, base seed {base} ({origin})\n")
    last = {"text": "(none yet)"}

    def announce(text):
        last["text"] = text
        trace.write(text + "\n")
        trace.flush()
        os.fsync(trace.fileno())

    rng = np.random.default_rng(seed)
    t0, i = time.time(), 0
    try:
        while time.time() - t0 < budget:
            if leg == "sat_poly_binned":
                ok, what, _ = fz.one(eng, rng, i, seed, announce)   # (the tool keys its polygons by seed * 100000 + index)
            else:
                ok, what = fz.one(eng, rng, i, announce)
            assert ok, f"{leg}: differs from the oracle at seed {seed} (C2D_FUZZ_SEED={base}, {origin}), {last['text']}; {what}"
            i += 1
    finally:
        trace.write(f"# {i} configurations completed in {time.time() - t0:.1f} s\n")
        trace.close()
    with capsys.disabled():
        print(f"[fuzz] {leg}: {i} configurations in {time.time() - t0:.0f} s, 0 differences", flush=True)
    assert i >= 3, f"{leg}: only {i} configurations fit into {budget} s"
