#!/bin/bash
set -o pipefail
O=gpurun_out/r06s
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -2 $O/$name.txt | cut -c1-400; return $rc; }
step ncc_tests 500 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_renderer.py -q -x -m gpu || exit 1
step headline_1 200 bash tools/quick_headline.sh
step headline_2 200 bash tools/quick_headline.sh
echo "END" | tee -a $O/steps.txt
