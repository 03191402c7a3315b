#!/bin/bash
# headline only: host threads x route x stagger (A/B of fb_match_strips against the numpy statement)
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 24 --warmup 3"
out=gpurun_out/sweep_threads.txt; : > $out
run() { echo "threads=$1 route=$2 stagger=$3" >> $out
  FEABAS_BENCH_STAGGER_MS=$3 FEABAS_HIP_STRIP_ROUTE=$2 python bench.py $F --host-threads $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])" >> $out || exit 1; }
run 4 native 0; run 4 native 5; run 4 host 0; run 8 native 0; run 8 native 3; run 8 host 0; run 3 native 0; run 3 native 7; run 2 native 10
cat $out
