#!/bin/bash
# headline only, with the per-kernel times of the one-stream roofline pass
F="--no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 24 --warmup 3"
python bench.py $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()}, d['roofline']['per_kernel_gbs'])"
