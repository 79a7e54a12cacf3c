"""Where the exploring legs of the differential fuzzers take their seed from (tests/test_gpu_fuzz_explore.py).

The fixed-seed legs of the suite replay the same configurations in every run: after their first green run they explore
nothing.  The exploring legs draw a seed that changes with every commit, so that every run of the suite at a new commit
(the builder's own, and the round-end driver's on a box the builder never sees) covers configurations no run has covered.
In order:
  1. $C2D_FUZZ_SEED                      — to reproduce a reported failure,
  2. `git rev-parse HEAD` of this tree   — the commit under test,
  3. build/commit_stamp.txt              — that commit as the last pytest / build() run with git at hand recorded it: the GPU box
                                           receives a snapshot of the tree WITHOUT .git, and build/ travels with it,
  4. a digest of the shipped kernel sources — never a constant.
TEST INFRASTRUCTURE."""
from __future__ import annotations

import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAMP = os.path.join(ROOT, "build", "commit_stamp.txt")


def git_head():
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=20)
    except (OSError, subprocess.TimeoutExpired):
        return None
    sha = out.stdout.strip()
    return sha if out.returncode == 0 and len(sha) >= 12 and all(c in "0123456789abcdef" for c in sha) else None


def write_stamp() -> None:
    """Record the commit under test where a snapshot without .git still finds it (called from tests/conftest.py and build())."""
    sha = git_head()
    if sha is None:
        return
    try:
        os.makedirs(os.path.dirname(STAMP), exist_ok=True)
        with open(STAMP, "w") as f:
            f.write(sha + "\n")
    except OSError:
        pass


def source_digest() -> str:
    h = hashlib.sha256()
    base = os.path.join(ROOT, "convex-2d-gpu-collision-detection_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(base, name), "rb").read())
    return h.hexdigest()


def commit_seed():
    """(seed < 2^31, where it came from)"""
    env = os.environ.get("C2D_FUZZ_SEED")
    if env:
        return int(env) & 0x7FFFFFFF, "$C2D_FUZZ_SEED"
    sha = git_head()
    origin = "git HEAD %s" % sha[:12] if sha else None
    if sha is None and os.path.exists(STAMP):
        sha = open(STAMP).read().strip() or None
        origin = "build/commit_stamp.txt %s" % sha[:12] if sha else None
    if sha is None:
        sha = source_digest()
        origin = "sha256 of csrc/*.hip, *.hpp %s (no git, no stamp)" % sha[:12]
    return int(sha[:12], 16) & 0x7FFFFFFF, origin
