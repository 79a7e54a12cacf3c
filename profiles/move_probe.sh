#!/bin/bash
# Developer probe of the binning pass (poly_bin_count / scan / move) on the GPU box:  bash profiles/move_probe.sh <tag> [granularity]
# kernel durations (kernel trace), then HBM bytes and SQ / LDS counters in separate --pmc passes (no tracing beside them).
set -o pipefail
TAG=${1:?tag}
G=${2:-1}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
P="python3 $R/tests/tools/binning_probe.py 10000000 $G"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $P > $O/probe.txt 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > /dev/null 2> $O/pmc_fetch.err; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > /dev/null 2> $O/pmc_write.err; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $P > /dev/null 2> $O/pmc_sq.err; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_lds -- $P > /dev/null 2> $O/pmc_lds.err; echo "lds rc=$?"
cd $R
{
  echo "== kernel durations (ns)"; grep -h "poly_bin" $O/stats/*/*kernel_stats.csv 2>/dev/null || grep -rh "poly_bin" $O/stats --include=*stats*.csv | head
  for d in pmc_fetch pmc_write pmc_sq pmc_lds; do echo "== $d"; python3 profiles/pmc_digest.py $O/$d | grep "poly_bin"; done
} > $O/move_probe.txt
find $O -name "*.csv" -size +2M -exec gzip -f {} \;
cat $O/move_probe.txt
