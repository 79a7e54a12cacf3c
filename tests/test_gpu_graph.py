"""The asynchronous entry points do no allocation and no synchronisation, so a caller can
capture them into a HIP graph (include/c2d.h conventions).  tests/graph_capture_check.py captures
rects_from_poses, sat_rect_pairs_verts, mc_pair and mc_scenes with torch's graph capture
(hipStreamBeginCapture on the stream handed to the C-ABI), replays the graph three times and
compares with the oracle.  It runs in its own process because torch has to be imported first."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_entry_points_are_graph_capturable():
    out = subprocess.run([sys.executable, os.path.join(HERE, "graph_capture_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "graph capture ok" in out.stdout
