#!/bin/bash
# Round 5 — every differential fuzzer once more at the final build (the count kernels carry a ticket now, the polygon Monte-Carlo
# has sixteen tame instances), new seeds.   bash profiles/r05_fuzz_soak.sh > profiles/r05_final_fuzz_soak.txt
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
T=tests/tools
echo "# final soak of round 5: every differential fuzzer at the final build, seed 5 (OpenMP oracle on $OMP_NUM_THREADS threads)"
for job in "mc_fuzz 600 5" "mc_poly_fuzz 400 5" "poly_fuzz 400 5" "pose_fuzz 600 5" "binned_fuzz 300 5" "mc_mixed_scale_fuzz 5 400" "pair_mixed_scale_fuzz 5"; do
  set -- $job
  echo "## $1"
  timeout -k 10 280 python3 $T/$1.py ${@:2} 2>&1 | tail -4
done
