#!/bin/bash
# headline only: sub-batch size of the streaming NCC class (Infinity Cache residency of T and V) x host threads
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 24 --warmup 3"
out=gpurun_out/sweep_arena.txt; : > $out
run() { echo "threads=$1 arena_mb=$2" >> $out
  FEABAS_HIP_NCC_ARENA_MB=$2 python bench.py $F --host-threads $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()})" >> $out || exit 1; }
run 1 8192; run 1 192; run 1 96; run 2 96; run 4 48; run 8 8192; run 8 512; run 8 64
cat $out
