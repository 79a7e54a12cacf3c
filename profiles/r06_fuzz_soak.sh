#!/bin/bash
# Round 6 — soak of the exploring fuzz legs (tests/test_gpu_fuzz_explore.py) over base seeds no run has used, each leg for
# $SECONDS_PER_LEG (default 150 s), plus the CLI fuzzers the exploring legs do not cover (mixed scales, polygon Monte-Carlo).
# Every leg prints its seed before it starts and names each configuration before it runs in gpurun_out/fuzz_trace/.
#   SEEDS="11 12 13" SECONDS_PER_LEG=150 bash profiles/r06_fuzz_soak.sh > profiles/r06_fuzz_soak.txt
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
T=tests/tools
echo "# soak of round 6: exploring fuzz legs, ${SECONDS_PER_LEG:-150} s per leg, base seeds ${SEEDS:-601 602} (OpenMP oracle on $OMP_NUM_THREADS threads)"
for S in ${SEEDS:-601 602}; do
  echo "## base seed $S"
  C2D_FUZZ_SEED=$S C2D_FUZZ_SECONDS=${SECONDS_PER_LEG:-150} timeout -k 10 ${LIMIT:-900} python3 -m pytest tests/test_gpu_fuzz_explore.py -m gpu -q -x 2>&1 | grep -E "^\[fuzz\]|passed|failed|Error|error" 
  rc=${PIPESTATUS[0]}
  [ $rc -ne 0 ] && { echo "pytest rc=$rc at base seed $S: stopping (no further GPU step after a failure)"; exit $rc; }
  for job in "mc_mixed_scale_fuzz $S 300" "pair_mixed_scale_fuzz $S" "mc_poly_fuzz 200 $S"; do
    set -- $job
    echo "### $1 ${@:2}"
    timeout -k 10 ${LIMIT:-280} python3 $T/$1.py ${@:2} 2>&1 | tail -2
    rc=${PIPESTATUS[0]}
    [ $rc -ne 0 ] && { echo "$1 rc=$rc at seed $S: stopping"; exit $rc; }
  done
  # the binning pass on its index-checked build (make lib-movecheck): every index the move kernel forms is compared with its array's
  # size and the first offender reported on stderr instead of being used — reads past an input cannot be seen from results
  echo "### binned_fuzz 600 $S on libc2d_movecheck.so"
  C2D_LIBRARY=$PWD/convex-2d-gpu-collision-detection_amd/lib/libc2d_movecheck.so timeout -k 10 ${LIMIT:-280} python3 $T/binned_fuzz.py 600 $S > /tmp/movecheck_$S.log 2>&1
  rc=$?
  tail -1 /tmp/movecheck_$S.log
  echo "index reports: $(grep -c 'c2d move check' /tmp/movecheck_$S.log)"
  [ $rc -ne 0 ] && { echo "binned_fuzz on the index-checked build rc=$rc at seed $S: stopping"; exit $rc; }
  grep -q 'c2d move check' /tmp/movecheck_$S.log && { echo "the index-checked build reported an access outside its arrays at seed $S: stopping"; grep 'c2d move check' /tmp/movecheck_$S.log | head -5; exit 9; }
done
echo "# done: 0 differences"
