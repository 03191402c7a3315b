#!/bin/bash
# First GPU call of the next round: the whole state of HEAD that has not run on hardware since the pool closed in round 5, in an
# order that cannot take a box down twice -- every step under its own timeout, steps joined by && (a killed step ends the call),
# output to gpurun_out/r06/.   usage: gpurun --timeout 1100 -- 'bash tools/r06_first.sh'
set -o pipefail
O=gpurun_out/r06
mkdir -p $O
(nproc; free -g; ulimit -a) > $O/box.txt 2>&1
step() {          # step <name> <seconds> <command...>
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $O/steps.txt
  timeout -k 10 $secs "$@" > $O/$name.txt 2>&1
  local rc=$?
  echo "   rc $rc" | tee -a $O/steps.txt
  tail -3 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi
  return $rc
}
step ncc 300 python -m pytest tests/test_gpu_ncc.py -q -m gpu -x &&
step fem 400 python -m pytest tests/test_gpu_fem.py -q -m gpu -x &&
step comm 200 python -m pytest tests/test_gpu_comm.py tests/test_dist_gloo.py -q -m gpu -x &&
step pipeline 400 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x &&
step fullsize 400 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x &&
step renderer 400 python -m pytest tests/test_gpu_renderer.py -q -m gpu -x &&
step rest 400 python -m pytest tests -q -m gpu -x --ignore=tests/test_gpu_ncc.py --ignore=tests/test_gpu_fem.py --ignore=tests/test_gpu_comm.py --ignore=tests/test_dist_gloo.py --ignore=tests/test_gpu_pipeline.py --ignore=tests/test_gpu_fullsize.py --ignore=tests/test_gpu_renderer.py &&
step smoke 120 python __graft_entry__.py smoke &&
step bench 600 python bench.py &&
echo "ALL GREEN" | tee -a $O/steps.txt
# NOT part of this call (each ALONE in a call of its own, under a short timeout, after reading tools/pending/README):
#   FEABAS_TEST_PENDING=1 python -m pytest tests/test_gpu_ncc.py -q -k g20_xcorr_normalized        (new kernels of fb_ncc_batch_normalized)
#   FEABAS_TEST_PENDING=1 python -m pytest tests/test_gpu_fem.py -q -k g21_grouped
#   FEABAS_TEST_PENDING=1 python -m pytest tests/test_gpu_pipeline.py -q -k "ragged_batch_photometric or ragged_batch_with_masks"   (then: stitching_matcher_batch may stop routing masked / photometric pairs per shape)
#   FEABAS_TEST_PENDING=1 python -m pytest tests/test_gpu_renderer.py -q -k section_matcher_vs_oracle
#   the PCG best-iterate patch (tools/pending/pcg_best_iterate.patch)
