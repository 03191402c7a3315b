#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  timeout -k 10 400 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --align-sections 0 "$@" > gpurun_out/b12.json 2> gpurun_out/b12.err
  python3 - "$*" <<EOF
import json, sys
d = json.loads([l for l in open("gpurun_out/b12.json") if l.startswith("{")][-1])
print(sys.argv[1] or "(default)", "| headline", round(d["value"]), "stitch edge", round(d["stitch_sections"]["edge"]["pairs_per_s"]), "corner", round(d["stitch_sections"]["corner"]["pairs_per_s"]))
EOF
}
run
run --no-fem
run --host-ingest-pairs 0
run --no-align
run --no-deformed
run --no-fem --host-ingest-pairs 0 --no-align --no-deformed
