#!/bin/bash
set -o pipefail
O=gpurun_out/r06l
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=24
step ncc_renderer 500 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_renderer.py tests/test_gpu_fullsize.py -q -x -m gpu || exit 1
step prof_align_section 300 python tools/prof_align_section.py
step prof_deformed 300 python tools/prof_deformed.py
step xcorr_classes 300 python bench.py --no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 8 --warmup 2
echo "END" | tee -a $O/steps.txt
