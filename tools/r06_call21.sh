#!/bin/bash
set -o pipefail
O=gpurun_out/r06v
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -2 $O/$name.txt | cut -c1-400; return $rc; }
true
for k in 1 2 3; do
  step headline_${k}_nt 200 bash tools/quick_headline.sh
  FEABAS_HIP_LIB=ab/lib_nont.so step headline_${k}_plain 200 bash tools/quick_headline.sh
done
echo "END" | tee -a $O/steps.txt
