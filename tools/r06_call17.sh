#!/bin/bash
set -o pipefail
O=gpurun_out/r06q
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -3 $O/$name.txt | cut -c1-300; return $rc; }
step fem 500 python -m pytest tests/test_gpu_fem.py -q -x -m gpu || exit 1
step section_profile 300 python tools/bench_section_matcher.py --profile
echo "END" | tee -a $O/steps.txt
