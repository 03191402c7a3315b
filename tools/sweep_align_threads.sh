#!/bin/bash
# config[4] per-rank workload against the number of host threads (one context + SLM each)
mkdir -p gpurun_out/g5
for T in ${THREADS:-1 2 3 4 6 8}; do
  python bench.py --steps 2 --warmup 1 --no-fem --no-align --no-cpu-baseline --no-xcorr-classes --no-deformed --host-ingest-pairs 0 --stitch-sections 0 --align-threads $T --align-sections ${SECTIONS:-32} > gpurun_out/g5/a$T.json 2> gpurun_out/g5/a$T.err || exit 1
  python - <<PY
import json
d = json.loads(open('gpurun_out/g5/a$T.json').read().strip().splitlines()[-1])
a = d['align_sections']
print($T, 'threads:', round(a['sections_per_s'], 1), 'sections/s; optimize_linear thread-seconds', round(a['optimize_linear_seconds_this_rank'], 3), 'of wall', round(a['seconds'], 3), 'first / median', a['optimize_linear_s_first_of_a_thread_and_median'])
PY
done
