#!/bin/bash
set -o pipefail
O=gpurun_out/r06g
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=16
step hbm_store_probe 200 ./ab/hbm_store_probe
step section_entries 300 python tools/bench_section_matcher.py --entries --profile
echo "END" | tee -a $O/steps.txt
