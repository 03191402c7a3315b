#!/bin/bash
# the driver's round-end sequence in one call: the whole -m gpu suite in ONE process, smoke, the default bench
set -o pipefail
O=gpurun_out/${1:-r06full}
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -3 $O/$name.txt | cut -c1-300
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
step gpu_suite 900 python -m pytest tests -x -q -m gpu &&
step smoke 200 python __graft_entry__.py smoke &&
step bench 600 python bench.py &&
echo "ALL GREEN" | tee -a $O/steps.txt
