#!/bin/bash
# Round 6 — BASELINE configs 2, 3, 4 and 5 AT THEIR STATED SIZE over further seeds than the suite's one (2 and 5: 1e7 pairs, every boolean
# against the oracle, plus the size-independent properties; 3: 1e8 samples hit for hit; 4: one GPU's 4e6-data-point shard — the two-halves
# identity, 20 000 data points against the oracle, the whole shard against the full-evaluation build): tests/test_gpu_sat.py::test_full_size_1e7_oracle_equality_and_properties and
# tests/fullsize_poly_check.py with $C2D_FULLSIZE_SEED.   SEEDS="21 22 23" bash profiles/r06_fullsize_seeds.sh > profiles/r06_fullsize_seeds.txt
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
echo "# configs 2, 5 (1e7 pairs, all booleans), 3 (1e8 samples, hit for hit) and 4 (4e6-data-point shard) against the oracle, seeds ${SEEDS:-21 22 23 24 25}"
for S in ${SEEDS:-21 22 23 24 25}; do
  echo "## seed $S"
  C2D_FULLSIZE_SEED=$S timeout -k 10 300 python3 -m pytest tests/test_gpu_sat.py -m gpu -q -x -k test_full_size_1e7_oracle_equality_and_properties 2>&1 | tail -1
  [ ${PIPESTATUS[0]} -ne 0 ] && { echo "config 2 failed at seed $S: stopping"; exit 1; }
  C2D_FULLSIZE_SEED=$S timeout -k 10 300 python3 tests/fullsize_poly_check.py 2>&1 | tail -1
  [ ${PIPESTATUS[0]} -ne 0 ] && { echo "config 5 failed at seed $S: stopping"; exit 1; }
  C2D_FULLSIZE_SEED=$S timeout -k 10 300 python3 -m pytest tests/test_gpu_mc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "test_mc_pair_1e8_vs_oracle_exact or test_config4_shard_at_full_size" 2>&1 | tail -1
  [ ${PIPESTATUS[0]} -ne 0 ] && { echo "config 3 / 4 failed at seed $S: stopping"; exit 1; }
done
echo "# done"
