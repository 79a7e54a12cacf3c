#!/bin/bash
# Round 5 — every differential fuzzer once more at the final build (the count kernels carry a ticket now, the polygon Monte-Carlo
# has sixteen tame instances), new seeds.   bash profiles/r05_fuzz_soak.sh > profiles/r05_final_fuzz_soak.txt
# SEED=<n> picks another seed (default 5), SCALE=<k> multiplies the configuration counts (default 1).
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
T=tests/tools
S=${SEED:-5}; K=${SCALE:-1}
echo "# soak of round 5: every differential fuzzer at the final build, seed $S (OpenMP oracle on $OMP_NUM_THREADS threads)"
for job in "mc_fuzz $((600 * K)) $S" "mc_poly_fuzz $((400 * K)) $S" "poly_fuzz $((400 * K)) $S" "pose_fuzz $((600 * K)) $S" "binned_fuzz $((300 * K)) $S" "mc_mixed_scale_fuzz $S $((400 * K))" "pair_mixed_scale_fuzz $S"; do
  set -- $job
  echo "## $1"
  timeout -k 10 ${LIMIT:-280} python3 $T/$1.py ${@:2} 2>&1 | tail -4
done
