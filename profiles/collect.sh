#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box:  bash profiles/collect.sh <tag>   (run from the repo root)
# Raw output goes to gpurun_out/<tag>/ (scratch); profiles/summarize.py + pmc_digest.py condense it into profiles/<tag>_*.
# Counter passes are separate runs without any tracing option, as the pool requires; the program after `--` is python3 itself.
set -e -o pipefail
TAG=${1:?tag}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
SMALL="--steps 20 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --mc-reps 2 --poly-reps 3 --poly-scenes 0"
cd /tmp
echo "== un-profiled bench"; timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo "== kernel trace + stats"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
echo "== pmc FETCH_SIZE"; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $SMALL --no-mc --scenes 0 > /dev/null 2> $O/pmc_fetch.err
echo "== pmc WRITE_SIZE"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $SMALL --no-mc --scenes 0 > /dev/null 2> $O/pmc_write.err
echo "== pmc SQ"; timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $SMALL --poly-scenes 200000 > /dev/null 2> $O/pmc_sq.err
echo "== pmc SQ LDS (polygon kernel)"; timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_lds -- python3 $R/bench.py $SMALL --no-mc --no-pose --scenes 0 > /dev/null 2> $O/pmc_lds.err
# Issue-weighted VALU roof of the Monte-Carlo legs (profiles/valu_issue.py): the per-type instruction counters and the busy / scalar /
# LDS-wait counters on those legs, the probe binary under the per-type counters (which class each instruction is counted in), the
# probe's prices, then each leg's own mix as a probe stream (generated, compiled here, run)
TYPES="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
BUSY="SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
MCLEGS="--steps 20 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --mc-reps 2 --poly-scenes 200000 --no-pose --poly-pairs 0"
PROBE=$R/convex-2d-gpu-collision-detection_amd/csrc/tools/instr_probe
echo "== pmc VALU types (Monte-Carlo legs)"; timeout -k 10 400 rocprofv3 --pmc $TYPES --output-format csv -d $O/pmc_valu_types -- python3 $R/bench.py $MCLEGS > /dev/null 2> $O/pmc_valu_types.err
echo "== pmc VALU busy / SALU / LDS wait (Monte-Carlo legs)"; timeout -k 10 400 rocprofv3 --pmc $BUSY --output-format csv -d $O/pmc_valu_busy -- python3 $R/bench.py $MCLEGS > /dev/null 2> $O/pmc_valu_busy.err
echo "== pmc VALU types (the probe binary)"; timeout -k 10 300 rocprofv3 --pmc $TYPES --output-format csv -d $O/probe_types -- $PROBE > /dev/null 2> $O/probe_types.err
echo "== adaptive-loop trace (reference-default batch)"; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/scenes_trace -- python3 $R/convex-2d-gpu-collision-detection_amd/csrc/tools/scenes_trace.py run $O/scenes > $O/scenes_run.txt 2> $O/scenes.err
cd $R
echo "== pose probe, clock probe"; timeout -k 10 120 convex-2d-gpu-collision-detection_amd/csrc/tools/pose_probe > profiles/${TAG}_pose_probe.txt
timeout -k 10 120 convex-2d-gpu-collision-detection_amd/csrc/tools/clock_probe > profiles/${TAG}_clock_probe_raw.txt
python3 profiles/summarize.py $TAG $O/stats $O/pmc_fetch $O/pmc_write
python3 profiles/pmc_digest.py $O/pmc_sq > profiles/${TAG}_pmc_sq.txt
python3 profiles/pmc_digest.py $O/pmc_lds > profiles/${TAG}_pmc_lds.txt
python3 convex-2d-gpu-collision-detection_amd/csrc/tools/scenes_trace.py digest $O/scenes_trace $O/scenes > profiles/${TAG}_scenes_trace.md
cp $O/bench.json profiles/${TAG}_bench.json
cp $O/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
python3 profiles/pmc_digest.py --wide $O/pmc_valu_types > profiles/${TAG}_pmc_valu_types.txt
python3 profiles/pmc_digest.py --wide $O/pmc_valu_busy > profiles/${TAG}_pmc_valu_busy.txt
python3 profiles/pmc_digest.py --wide --all $O/probe_types > profiles/${TAG}_probe_types.txt
echo "== instruction probe: prices, then the legs' own mixes"
timeout -k 10 200 $PROBE > profiles/${TAG}_instr_probe.txt                 # prices (with whatever mixes the tree's probe holds)
python3 profiles/valu_issue.py mixes $TAG                                  # csrc/tools/instr_probe_mixes.inc from THIS tag's counts
make tools > $O/make_tools.log 2>&1 || echo "probe rebuild failed"
timeout -k 10 200 $PROBE > profiles/${TAG}_instr_probe.txt                 # prices and the rates of this tag's mixes
python3 profiles/valu_issue.py write $TAG && python3 profiles/valu_issue.py verify $TAG
cp convex-2d-gpu-collision-detection_amd/csrc/tools/instr_probe_mixes.inc $R/gpurun_out/${TAG}_instr_probe_mixes.inc   # (generated source: travels back through gpurun_out/)
python3 profiles/counts.py write $TAG   # measured_counts.json is generated from this tag's digests (tests/test_profiles.py verifies it)
# fields other builds record into it: the census build's evaluated-sample fraction, the clock build's held clock (make lib-mcstats lib-mcclock)
if [ -f convex-2d-gpu-collision-detection_amd/lib/libc2d_mcstats.so ]; then echo "== census"; timeout -k 10 400 python3 tests/tools/mc_stats.py --record > profiles/${TAG}_mc_stats.txt 2>&1 || echo "census failed"; fi
if [ -f convex-2d-gpu-collision-detection_amd/lib/libc2d_mcclock.so ]; then echo "== held clock"; timeout -k 10 400 python3 tests/tools/mc_clock.py --record $TAG > profiles/${TAG}_mc_clock.txt 2>&1 || echo "clock failed"; fi
# the bench line quotes measured_counts.json (instruction counts, PMC traffic): run it again now that the file is this tag's own,
# so that the committed line never carries the previous collection's counts for a kernel that has changed since
echo "== un-profiled bench again, with this tag's counts"; (cd /tmp && timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err)
cp $O/bench.json profiles/${TAG}_bench.json
mkdir -p $R/gpurun_out/${TAG}_profiles && cp profiles/${TAG}_* profiles/measured_counts.json $R/gpurun_out/${TAG}_profiles/   # (only gpurun_out/ travels back from the GPU box)
# the raw traces are large: keep only what the digests came from, compressed
find $O -name "*.csv" -size +2M -exec gzip -f {} \;
find $O -name "*.gz" -size +20M -delete
echo "== done"
