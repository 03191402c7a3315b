#!/bin/bash
# headline only: hardware queues of the HIP runtime x host threads x sub-batch
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 24 --warmup 3"
out=gpurun_out/sweep_queues.txt; : > $out
run() { echo "hwq=$1 threads=$2 sub=$3 stagger=$4" >> $out
  if [ "$1" = "default" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$1; fi
  FEABAS_BENCH_STAGGER_MS=$4 python bench.py $F --host-threads $2 --sub-batch $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])" >> $out || exit 1; }
run default 8 128 3; run 8 8 128 3; run 16 8 128 3; run 8 12 128 2; run 16 16 64 1; run 8 8 64 2; run 2 8 128 3; run default 8 128 3
cat $out
