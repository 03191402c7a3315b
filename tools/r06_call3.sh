#!/bin/bash
# call 3 of round 6: the pending pipeline tests (masks / photometric statistics in ragged batches), then a8 end to end (section 0 locked), alone
set -o pipefail
O=gpurun_out/r06c
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_TEST_PENDING=1
export FEABAS_RSS_LIMIT_GB=16
step ragged_pending 300 python -m pytest tests/test_gpu_pipeline.py -q -x -k "ragged_batch_photometric or ragged_batch_with_masks" &&
FEABAS_HIP_PCG_TRACE=1 step section_locked 120 python -m pytest tests/test_gpu_renderer.py -q -x -s -k section_matcher_vs_oracle &&
echo "ALL GREEN" | tee -a $O/steps.txt
