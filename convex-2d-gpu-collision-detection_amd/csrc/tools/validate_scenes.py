"""One-off soundness check of the Monte-Carlo pretests on the adaptive path: per-scene hit and sample counts of 2e6
scenes (shape variance on) from the shipped build vs the full-evaluation build (make lib-nopretest) must have the same
SHA-256.  Developer tool, GPU only.  Round 1: identical (2.03e11 samples)."""
import sys, os, json, subprocess, hashlib
code = r'''
import sys, os, json, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, importlib
from __graft_entry__ import load_package
pkg = load_package(); wl = importlib.import_module("c2d_amd.workloads"); eng = pkg.Engine(0)
ns, npose = 2_000_000, 65536
poses, sds, _ = wl.random_tables(npose, npose, seed=17, shape_variance=True)
d_p, d_s = eng.to_device(poses), eng.to_device(sds)
d_sc = eng.empty(ns, pkg.SCENE_DT)
eng.sample_scenes(d_p, npose, d_s, npose, 4.07, 1.74, 4.0, 23, 0, ns, d_sc)
d_h, d_u = eng.zeros(ns, np.uint32), eng.zeros(ns, np.uint32)
total, iters = eng.mc_scenes(d_p, npose, d_s, npose, d_sc, ns, 4.07, 1.74, wl.DEFAULT_BINS, wl.DEFAULT_BIN_ACCURACY, 120000, 31, 0, d_h, d_u, None)
h, u = d_h.get(), d_u.get()
print(json.dumps({"total": total, "iters": iters, "hits_sum": int(h.astype(np.int64).sum()), "sha_hits": hashlib.sha256(h.tobytes()).hexdigest(), "sha_used": hashlib.sha256(u.tobytes()).hexdigest()}))
'''
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
res = {}
for name, lib in (("shipped", "convex-2d-gpu-collision-detection_amd/lib/libc2d.so"), ("full", "convex-2d-gpu-collision-detection_amd/lib/libc2d_nopretest.so")):
    env = dict(os.environ, C2D_LIBRARY=os.path.join(root, lib))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    res[name] = json.loads(r.stdout.strip().splitlines()[-1]); print(name, res[name])
print("identical:", res["shipped"] == res["full"])
sys.exit(0 if res["shipped"] == res["full"] else 1)
