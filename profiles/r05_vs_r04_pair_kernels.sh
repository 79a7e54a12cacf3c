#!/bin/bash
# Round 5 — did the workspace stamps (a ticket in the kernel arguments, one atomic per completed count word) cost the pair kernels
# anything?  The round-4 tree (git archive 906a4e7 -> build/r4tree, `make lib` there) and this tree, bench.py's pair legs, two
# rounds each on ONE box; HIP-event kernel times of every leg.   bash profiles/r05_vs_r04_pair_kernels.sh > out.txt
F="--no-cpu-baseline --no-mc --scenes 0 --steps 200 --warmup 20"
for round in 1 2; do
  for tree in build/r4tree .; do
    (cd $tree && timeout -k 10 200 python3 bench.py $F 2>/dev/null) | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = j['poly']
print('%-14s verts %.2f  mask %.2f  pose %.2f  poly16 %.2f  binned %.2f  poly4 %.2f  (us per 1e7 pairs)' % ('$tree', j['roofline']['kernel_ms'] * 1e3, j['mask_output']['kernel_ms'] * 1e3, j['pose_format']['kernel_ms'] * 1e3, p['kernel_ms'] * 1e3, p['binned']['kernel_ms'] * 1e3, p['small_polygons']['kernel_ms'] * 1e3))"
  done
done
