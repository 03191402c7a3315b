#!/bin/bash
# L2-miss read traffic (FETCH_SIZE) of the two PCG kernels at 1e6 DoF with the per-XCD row mapping on and off
# usage: bash tools/pmc_pcg_xcd.sh tag
set -u
TAG=${1:-pcgxcd}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for X in 1 0; do
  export FEABAS_HIP_SPMV_XCD=$X
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/xcd$X -o f -- python3 tools/prof_pcg_iter.py > $OUT/xcd$X.log 2>&1 || exit 2
done
python3 - <<PY
import csv, glob, collections
for X in (1, 0):
    f = glob.glob('$OUT/xcd%d/**/*counter_collection.csv' % X, recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'FETCH_SIZE':
            continue
        k = r['Kernel_Name']
        name = 'pcg_spmv (bsr_spmv_kernel<1>)' if 'bsr_spmv_kernel<1>' in k else 'pcg_update' if 'pcg_update_kernel' in k else None
        if name:
            acc[name][0] += 1; acc[name][1] += float(r['Counter_Value'])
    for name, (n, v) in sorted(acc.items()):
        print('per-XCD rows %d  %-32s launches %5d  FETCH_SIZE x 2 KiB per launch %10.0f  = %7.1f MB read from HBM / Infinity Cache' % (X, name, n, 2 * v / n, 2 * v / n * 1024 / 1e6))
PY
