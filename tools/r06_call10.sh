#!/bin/bash
set -o pipefail
O=gpurun_out/r06j
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
step dog_tests 300 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_pipeline.py -q -x -m gpu -k "dog or DoG or pipeline or strip" || exit 1
k=0
for m in 0 1 0 1; do
  k=$((k+1))
  FEABAS_HIP_DOG_XCD=$m step headline_${k}_dogxcd$m 200 bash tools/quick_headline.sh
done
echo "END" | tee -a $O/steps.txt
