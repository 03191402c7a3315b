#!/bin/bash
# SQ counters of the on-chip NCC class and the DoG (two passes of 8 SQ slots each), micro-benchmarks only
# usage: bash tools/pmc_sq.sh r02a
set -u
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for tool in microbench_small microbench_dog; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${tool}_a -o a -- python3 tools/$tool.py > $OUT/${tool}_a.log 2>&1 || exit 2
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${tool}_b -o b -- python3 tools/$tool.py > $OUT/${tool}_b.log 2>&1 || exit 3
done
find $OUT -name "*counter_collection.csv"
