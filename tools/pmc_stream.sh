#!/bin/bash
# SQ counters of the streaming NCC kernels (coarse-block shape and global-strip shape), two passes of 8 SQ slots each
# usage: bash tools/pmc_stream.sh r02b
set -u
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for shape in "1024 510" "2048 255"; do
  set -- $shape
  export BH=$1 BW=$2 NB=128 REPS=3
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/stream_${BH}_a -o a -- python3 tools/microbench_stream.py > $OUT/stream_${BH}_a.log 2>&1 || exit 2
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/stream_${BH}_b -o b -- python3 tools/microbench_stream.py > $OUT/stream_${BH}_b.log 2>&1 || exit 3
  python3 tools/microbench_stream.py > $OUT/stream_${BH}_plain.log 2>&1
done
find $OUT -name "*counter_collection.csv"
