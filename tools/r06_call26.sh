#!/bin/bash
O=gpurun_out/r06fz
mkdir -p $O
export PYTHONPATH=$PWD
export FEABAS_RSS_LIMIT_GB=24
for job in "fuzz_native.py 5 24" "fuzz_native.py 11 24" "fuzz_batch.py" "fuzz_pairs.py" "fuzz_pairs.py big"; do
  n=$(echo $job | tr ' ./' '___')
  timeout -k 10 400 python tools/$job > $O/$n.txt 2>&1; echo "$job rc $? : $(tail -n 2 $O/$n.txt | tr '\n' ' ' | cut -c1-300)"
done
