#!/bin/bash
# call 2 of round 6: the device PCG on floating systems (solver only, nothing applied to a mesh), old leg rule then new; then the
# small pending device tests, each in a process of its own under its own timeout
set -o pipefail
O=gpurun_out/r06b
mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name" | tee -a $O/steps.txt; timeout -k 10 $secs "$@" > $O/$name.txt 2>&1; local rc=$?; echo "   rc $rc" | tee -a $O/steps.txt; tail -4 $O/$name.txt
  if grep -q "Memory access fault" $O/$name.txt; then echo "GPU FAULT in $name" | tee -a $O/steps.txt; return 99; fi; return $rc; }
export FEABAS_TEST_PENDING=1
step pcg_probe_old 300 python tools/probe_pcg_floating.py 0 &&
step pcg_probe_new 300 python tools/probe_pcg_floating.py 1 &&
step fem_all 400 python -m pytest tests/test_gpu_fem.py -q -m gpu -x &&
step g20 180 python -m pytest tests/test_gpu_ncc.py -q -k g20_xcorr_normalized &&
echo "ALL GREEN" | tee -a $O/steps.txt
