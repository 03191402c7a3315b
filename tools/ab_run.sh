#!/bin/bash
# runs a command once per A/B library: bash tools/ab_run.sh "<command>" lib1.so lib2.so ...   (output to gpurun_out/ab_<lib>.log)
CMD=$1; shift
mkdir -p gpurun_out
for L in "$@"; do
  n=$(basename $L .so)
  echo "=== $n"
  FEABAS_HIP_LIB=$L timeout -k 10 300 bash -c "$CMD" > gpurun_out/ab_$n.log 2>&1 || echo "FAILED $n"
  tail -8 gpurun_out/ab_$n.log
done
