#!/bin/bash
# call 5 of round 6: the floating section test, the renderer file, the files touched by the hook split; then where the host time of
# section_matcher goes and the ingest trace
set -o pipefail
O=gpurun_out/r06e
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=16
step section_floating 200 python -m pytest tests/test_gpu_renderer.py -q -x -k section_matcher_floating_pair &&
step renderer_all 400 python -m pytest tests/test_gpu_renderer.py -q -x &&
step fft_core_and_host 300 python -m pytest tests/test_gpu_fft_core.py tests/test_cpu_host.py -q -x &&
step pipeline 400 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x &&
step section_entries 300 python tools/bench_section_matcher.py --entries --profile
FEABAS_HIP_INGEST_TRACE=1 step ingest_trace 300 python bench.py --no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --no-xcorr-classes --steps 8 --warmup 2
echo "END" | tee -a $O/steps.txt
