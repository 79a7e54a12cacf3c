"""Maximum sizes: a 1.2e9-pair batch (planes of 4.8 GB, offsets beyond 2^32 bytes, 4.7e6 blocks) and
Monte-Carlo sample indices beyond 2^32.  The work is in tests/large_size_check.py, run in its own
process because it builds its inputs with torch (which has to be imported before libc2d.so)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_maximum_sizes():
    out = subprocess.run([sys.executable, os.path.join(HERE, "large_size_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "large size ok" in out.stdout
