#!/bin/bash
# Round 5 — what prices a VALU instruction on gfx950: csrc/tools/instr_probe alone (issue ticks per wave instruction), the same
# binary under the per-type instruction counters (which hardware class each instruction is counted in) and under the busy
# counters, then the Monte-Carlo legs of bench.py under both counter sets.  Counter passes are separate runs without tracing.
# bash profiles/r05_valu_probe.sh [out-dir-name]
set -o pipefail
R=$PWD; O=$R/gpurun_out/${1:-r5e}; mkdir -p $O; export TMPDIR=/tmp
T=convex-2d-gpu-collision-detection_amd/csrc/tools
timeout -k 10 120 $T/instr_probe > $O/instr_probe.txt 2>&1; echo probe rc=$?
cd /tmp
TYPES="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
BUSY="SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $TYPES --output-format csv -d $O/probe_types -- $R/$T/instr_probe > $O/probe_types.out 2>&1; echo probe types rc=$?
timeout -k 10 200 rocprofv3 --pmc $BUSY --output-format csv -d $O/probe_busy -- $R/$T/instr_probe > $O/probe_busy.out 2>&1; echo probe busy rc=$?
SMALL="--steps 20 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --mc-reps 2 --poly-reps 3 --poly-scenes 200000 --no-pose --poly-pairs 0"
timeout -k 10 400 rocprofv3 --pmc $TYPES --output-format csv -d $O/mc_types -- python3 $R/bench.py $SMALL > /dev/null 2> $O/mc_types.err; echo mc types rc=$?
timeout -k 10 400 rocprofv3 --pmc $BUSY --output-format csv -d $O/mc_busy -- python3 $R/bench.py $SMALL > /dev/null 2> $O/mc_busy.err; echo mc busy rc=$?
cd $R
for d in probe_types probe_busy mc_types mc_busy; do python3 - $O/$d > $O/$d.txt <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
acc = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:50s} {c:30s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
done
find $O -name "*.csv" -size +1M -delete
