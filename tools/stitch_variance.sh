#!/bin/bash
# the stitch_sections sub-record three times in a row on one box (run-to-run spread of the sharded workload)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-fem --no-align --host-ingest-pairs 0 --no-deformed --no-cpu-baseline --align-sections 0 > gpurun_out/b11_$i.json 2> gpurun_out/b11.err
  python3 - <<EOF
import json
d = json.loads([l for l in open("gpurun_out/b11_$i.json") if l.startswith("{")][-1])
print(d["value"], d["stitch_sections"]["edge"]["pairs_per_s"], d["stitch_sections"]["corner"]["pairs_per_s"])
EOF
done
