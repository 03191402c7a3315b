#!/bin/bash
# Collects what profiles/ holds for one round on the GPU box: the default bench line, rocprofv3 kernel stats of the
# same command (multi-stream and one-stream), and the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs).
# usage: bash tools/profile_round.sh r01g
set -u
TAG=${1:-r}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ms -o ms -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof_ms.err || exit 2
echo "stats (multi-stream) done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_os -o os -- python3 bench.py --no-cpu-baseline --no-fem --host-ingest-pairs 0 --no-align --no-deformed --stitch-sections 0 --align-sections 0 --multi-stream 0 > $OUT/bench_one_stream_under_rocprof.json 2> $OUT/prof_os.err || exit 3
echo "stats (one stream) done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --no-fem --host-ingest-pairs 0 --no-align --no-deformed --stitch-sections 0 --align-sections 0 --multi-stream 0 --steps 2 --warmup 0 --no-xcorr-classes > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 4
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --no-cpu-baseline --no-fem --host-ingest-pairs 0 --no-align --no-deformed --stitch-sections 0 --align-sections 0 --multi-stream 0 --steps 2 --warmup 0 --no-xcorr-classes > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 5
echo "pmc write done"
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
python3 tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1 || exit 6
echo "pmc traffic done"
# issue-rate probe of the VALU forms the DoG / FFT kernels lean on (tools/valu_rate.hip; DESIGN sec.9 quotes its table)
mkdir -p $ROOT/ab && hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o $ROOT/ab/valu_rate 2> $OUT/valu_rate.err && timeout -k 10 120 $ROOT/ab/valu_rate > $OUT/valu_rate.txt 2>> $OUT/valu_rate.err || exit 7
echo "valu probe done"
