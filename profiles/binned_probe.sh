#!/bin/bash
# Developer probe of the binned polygon kernel on the GPU box:  bash profiles/binned_probe.sh <tag> [granularities]
# timing (tests/tools/binned_bench.py), then SQ / LDS counters of the same run (separate --pmc passes, no tracing).
set -o pipefail
TAG=${1:?tag}
G=${2:-1}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $R/tests/tools/binned_bench.py 10000000 20 3 16 8.0 $G > $O/binned_bench.txt 2>&1; echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 $R/tests/tools/binned_bench.py 10000000 3 3 16 8.0 ${G%%,*} > /dev/null 2> $O/pmc_sq.err; echo "pmc sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_lds -- python3 $R/tests/tools/binned_bench.py 10000000 3 3 16 8.0 ${G%%,*} > /dev/null 2> $O/pmc_lds.err; echo "pmc lds rc=$?"
cd $R
python3 profiles/pmc_digest.py $O/pmc_sq > $O/pmc_sq.txt
python3 profiles/pmc_digest.py $O/pmc_lds > $O/pmc_lds.txt
find $O -name "*.csv" -size +2M -exec gzip -f {} \;
cat $O/binned_bench.txt; grep "binned\|sat_poly_kernel" $O/pmc_sq.txt $O/pmc_lds.txt
