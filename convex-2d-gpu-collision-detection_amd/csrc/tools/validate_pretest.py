"""One-off soundness check of the Monte-Carlo pretests: hit counts of the shipped build vs a build that
evaluates every sample in full (make lib-nopretest: -DC2D_MC_NO_PRETEST), on random scenes spread around
the pretest boundary.  Developer tool, GPU only:  python validate_pretest.py <samples per scene> <scenes>
Round 1: 2000 scenes x 1e9 samples (p from 0 to 1, median 6.6e-4, 365 zero-hit scenes): identical."""
import sys, os, json, subprocess
code = r'''
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); eng = pkg.Engine(0)
rng = np.random.default_rng(2025)
res = []
N = int(os.environ["NSAMP"])
for i in range(int(os.environ["NSCENES"])):
    w, h = rng.uniform(0.1, 5, 2); th = rng.uniform(0, 6.283)
    sd = tuple(np.sqrt(rng.uniform(0, 0.3, 3)).tolist()) + ((float(np.sqrt(rng.uniform(0, 0.3))), float(np.sqrt(rng.uniform(0, 0.3)))) if i % 3 == 0 else (0.0, 0.0))
    rho = np.hypot(w / 2 + 3.385 * sd[3], h / 2 + 3.385 * sd[4])
    # distance: from overlapping to far, concentrated where collisions become rare
    dist = rho + rng.choice([0.87, 2.035]) + rng.uniform(-1.0, 4.0) * max(sd[0], sd[1], 0.05)
    ang = rng.uniform(0, 6.283)
    pos = (float(dist * np.cos(ang)), float(dist * np.sin(ang)))
    d = eng.zeros(1, np.uint64)
    eng.mc_pair(4.07, 1.74, pos, (float(w), float(h), float(th)), sd, 777, i, 0, N, d)
    res.append(int(d.get()[0])); d.free()
print(json.dumps(res))
'''
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = {}
for name, lib in (("shipped", "convex-2d-gpu-collision-detection_amd/lib/libc2d.so"), ("full", "convex-2d-gpu-collision-detection_amd/lib/libc2d_nopretest.so")):
    env = dict(os.environ, C2D_LIBRARY=os.path.join(root, lib), NSAMP=sys.argv[1], NSCENES=sys.argv[2])
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    out[name] = json.loads(r.stdout.strip().splitlines()[-1])
a, b = out["shipped"], out["full"]
n = int(sys.argv[1])
print("scenes", len(a), "samples each", n, "identical:", a == b)
ps = sorted(x / n for x in b)
print("p quantiles:", [round(ps[int(q * (len(ps) - 1))], 6) for q in (0, .1, .25, .5, .75, .9, 1)], "zero-hit scenes:", sum(1 for x in b if x == 0))
if a != b:
    print([(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y][:10]); sys.exit(1)
