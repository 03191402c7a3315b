#!/bin/bash
# the `deformed` record of bench.py against the host-thread knobs
mkdir -p gpurun_out/g22
run() {
  python bench.py --steps 8 --warmup 2 --no-fem --no-align --no-cpu-baseline --no-xcorr-classes --host-ingest-pairs 0 --stitch-sections 0 --align-sections 0 "$@" 2> /dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value']), 'deformed', round(d['deformed']['value']))"
}
echo "default"; run
echo "FEABAS_HIP_HOST_THREADS=1"; FEABAS_HIP_HOST_THREADS=1 run
echo "--host-threads 12"; run --host-threads 12
echo "--host-threads 16"; run --host-threads 16
