#!/bin/bash
# kernel stats of the headline workload on one stream (rocprofv3 --kernel-trace --stats)
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --no-xcorr-classes --multi-stream 0 --steps 8 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py $F > $OUT/bench_one_stream.json 2> $OUT/bench_one_stream.err || exit 2
python3 tools/kernel_stats_top.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) 14
