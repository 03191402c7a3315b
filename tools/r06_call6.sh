#!/bin/bash
set -o pipefail
O=gpurun_out/r06f
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=16
step fft_core_and_host 300 python -m pytest tests/test_gpu_fft_core.py tests/test_cpu_host.py -q -x
step pipeline 400 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x
step section_entries 300 python tools/bench_section_matcher.py --entries --profile
echo "END" | tee -a $O/steps.txt
