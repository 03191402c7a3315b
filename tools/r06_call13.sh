#!/bin/bash
set -o pipefail
O=gpurun_out/r06m
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_RSS_LIMIT_GB=24
step pipeline 500 python -m pytest tests/test_gpu_pipeline.py -q -x -m gpu || exit 1
step ingest_split 400 python bench.py --no-fem --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --no-xcorr-classes --steps 8 --warmup 2
FEABAS_HIP_INGEST_NO_SPLIT=1 step ingest_nosplit 400 python bench.py --no-fem --no-align --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --no-xcorr-classes --steps 8 --warmup 2
echo "END" | tee -a $O/steps.txt
