#!/bin/bash
# stage trace of fb_match_strips under the headline workload, 1 and 4 host threads
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 24 --warmup 3"
out=gpurun_out/trace_native.txt; : > $out
for t in 1; do
  echo "threads=$t" >> $out
  FEABAS_HIP_MATCH_TRACE=1 python bench.py $F --host-threads $t 2>> $out | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])" >> $out || exit 1
done
cat $out
