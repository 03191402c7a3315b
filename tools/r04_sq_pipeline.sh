#!/bin/bash
# SQ counters (two passes of 8 slots) of every kernel of the headline pipeline, one stream, one step; summaries per kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04sq}
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
F="--no-cpu-baseline --no-fem --host-ingest-pairs 0 --no-align --no-deformed --stitch-sections 0 --align-sections 0 --multi-stream 0 --host-threads 1 --steps 1 --warmup 1 --no-xcorr-classes"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o a -- python3 bench.py $F > $OUT/a.json 2> $OUT/a.err || exit 2
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o b -- python3 bench.py $F > $OUT/b.json 2> $OUT/b.err || exit 3
python3 tools/pmc_summary.py $(find $OUT/a -name "*counter_collection.csv" | head -1) > $OUT/sq_a.txt
python3 tools/pmc_summary.py $(find $OUT/b -name "*counter_collection.csv" | head -1) > $OUT/sq_b.txt
grep -c . $OUT/sq_a.txt $OUT/sq_b.txt
