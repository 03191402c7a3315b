#!/bin/bash
set -o pipefail
O=gpurun_out/r06o
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -2 $O/$name.txt | cut -c1-400
  return $rc; }
k=0
for m in default 1 0 default 1 0; do
  k=$((k+1))
  if [ $m = default ]; then unset FB_INV_HALF; else export FB_INV_HALF=$m; fi
  step headline_${k}_invhalf_$m 200 bash tools/quick_headline.sh
done
echo "END" | tee -a $O/steps.txt
