#!/bin/bash
# SQ instruction counters of every kernel of the headline pipeline (one stream, one step): how busy is the VALU?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_valu
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
F="--no-cpu-baseline --no-fem --host-ingest-pairs 0 --no-align --no-deformed --stitch-sections 0 --align-sections 0 --multi-stream 0 --host-threads 1 --steps 1 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o a -- python3 bench.py $F > $OUT/a.json 2> $OUT/a.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 bench.py $F > $OUT/t.json 2> $OUT/t.err || exit 3
find $OUT -name "*.csv" | head
