#!/bin/bash
# Round 5 — A/B of wave priorities in sat_rect_verts_kernel<4, 64> (the headline kernel), against the pure stream of the same access
# pattern (csrc/tools/load_policy_probe).  build/prio1/libc2d.so: s_setprio 3 while a wave forms its addresses and issues its sixteen
# loads; build/prio2: s_setprio 3 once a wave's data has arrived (-DC2D_SAT_PRIO=1 / 2 on c2d_sat.hip; see the kernel).
# bash profiles/r05_sat_prio_ab.sh > profiles/r05_sat_prio_ab.txt     (same box, builds interleaved, twice)
LEG="--no-mc --scenes 0 --poly-pairs 0 --no-cpu-baseline"
timeout -k 10 200 convex-2d-gpu-collision-detection_amd/csrc/tools/load_policy_probe | grep "product\|again\|^#"
for round in 1 2; do
  for lib in convex-2d-gpu-collision-detection_amd/lib/libc2d.so build/prio1/libc2d.so build/prio2/libc2d.so; do
    [ -f $lib ] || continue
    C2D_LIBRARY=$PWD/$lib timeout -k 10 250 python3 bench.py $LEG 2>/dev/null | python3 -c "
import json, sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
r, m = b['roofline'], b.get('mask_output') or {}
print('%-28s verts: kernel %.2f us, frac %.4f, steps median %.2f min %.2f us | bit-mask output: %s' % ('$lib'.split('/')[-2], r['kernel_ms'] * 1e3, r['frac'], r['step_ms_distribution']['median'] * 1e3, r['step_ms_distribution']['min'] * 1e3, m.get('roofline', {}).get('frac')))"
  done
done
