#!/bin/bash
set -o pipefail
O=gpurun_out/r06h
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
step ncc_tests 300 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_fullsize.py -q -x -m gpu || exit 1
k=0
for m in 0 7 1 2 3 4 0 7; do
  k=$((k+1))
  FEABAS_HIP_P2_XCD=$m step headline_${k}_xcd$m 200 bash tools/quick_headline.sh
done
echo "END" | tee -a $O/steps.txt
