#!/bin/bash
# round 4: the on-chip 75 x 75 class before (ncc_small_fused) and after (ncc_pfa75): parity tests, micro-benchmark, SQ counters
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04small
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_ncc.py -x -q > $OUT/pytest_ncc.log 2>&1; echo "pytest ncc rc=$?"; tail -3 $OUT/pytest_ncc.log
python tools/microbench_small.py > $OUT/mb_new.txt 2>&1 && cat $OUT/mb_new.txt
FEABAS_HIP_NO_PFA=1 python tools/microbench_small.py > $OUT/mb_old.txt 2>&1 && cat $OUT/mb_old.txt
for v in new old; do
  if [ $v == old ]; then export FEABAS_HIP_NO_PFA=1; else unset FEABAS_HIP_NO_PFA; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${v}_a -o a -- python3 tools/microbench_small.py > $OUT/${v}_a.log 2>&1 || exit 2
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${v}_b -o b -- python3 tools/microbench_small.py > $OUT/${v}_b.log 2>&1 || exit 3
  python3 tools/pmc_summary.py $(find $OUT/${v}_a -name "*counter_collection.csv" | head -1) > $OUT/sq_${v}_a.txt
  python3 tools/pmc_summary.py $(find $OUT/${v}_b -name "*counter_collection.csv" | head -1) > $OUT/sq_${v}_b.txt
done
unset FEABAS_HIP_NO_PFA
grep -A9 "ncc_" $OUT/sq_new_a.txt $OUT/sq_old_a.txt | grep -v rocclr
