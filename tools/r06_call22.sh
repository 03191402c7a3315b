#!/bin/bash
set -o pipefail
O=gpurun_out/r06w
mkdir -p $O
export PYTHONPATH=$PWD
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -2 $O/$name.txt | cut -c1-300; return $rc; }
step tests 600 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_fullsize.py tests/test_gpu_renderer.py tests/test_gpu_fft_core.py -q -x -m gpu || exit 1
step bench_align 400 python bench.py --no-fem --no-cpu-baseline --no-deformed --stitch-sections 0 --align-sections 0 --host-ingest-pairs 0 --steps 8 --warmup 2
echo "END" | tee -a $O/steps.txt
