#!/bin/bash
# A/B of the streaming NCC kernels: per-kernel times at the two coarse shapes of the 4k pair, then the headline
for L in "$@"; do
  n=$(basename $L .so)
  echo "=== $n"
  for shp in "1024 510" "510 1024" "2048 255" "255 2048"; do
    set -- $shp
    FEABAS_HIP_LIB=$L NB=128 BH=$1 BW=$2 REPS=8 timeout -k 10 120 python tools/microbench_stream.py 2>&1 | grep -E "ncc_stream" | awk -v s="$1x$2" '{print s, $0}'
  done
  FEABAS_HIP_LIB=$L timeout -k 10 300 bash tools/quick_headline.sh 2>&1 | tail -1
done
