#!/bin/bash
# call 4 of round 6: the deflated PCG (probe, its two tests, the whole FEM file), then the floating-pair section test alone
set -o pipefail
O=gpurun_out/r06d
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=16
true
step pcg_tests 300 python -m pytest tests/test_gpu_fem.py -q -x -k "floating_systems_deflates or best_iterate" &&
step fem_all 400 python -m pytest tests/test_gpu_fem.py -q -m gpu -x &&
FEABAS_HIP_PCG_TRACE=1 step section_floating 200 python -m pytest tests/test_gpu_renderer.py -q -x -s -k section_matcher_floating_pair &&
step renderer_all 400 python -m pytest tests/test_gpu_renderer.py -q -x &&
echo "ALL GREEN" | tee -a $O/steps.txt
