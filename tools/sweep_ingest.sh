#!/bin/bash
# PCIe-inclusive sub-record: host threads x chunk size of stitching_matcher_batch
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --steps 4 --warmup 1"
out=gpurun_out/sweep_ingest.txt; : > $out
run() { echo "ingest threads=$1 batch=$2" >> $out
  python bench.py $F --host-ingest-threads $1 --host-ingest-batch $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['host_ingest']['value']), round(d['host_ingest']['ragged']['value']))" >> $out || exit 1; }
run 4 128; run 8 64; run 8 32; run 12 32; run 6 64; run 8 128
cat $out
