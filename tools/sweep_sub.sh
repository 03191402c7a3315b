#!/bin/bash
out=gpurun_out/sweep_sub.txt; : > $out
for a in "64 512" "128 512" "256 1024" "128 1024" "256 2048"; do
  set -- $a
  echo "sub=$1 pairs_per_step=$2" >> $out
  bash tools/quick_headline.sh --sub-batch $1 --pairs-per-step $2 --resident-pairs 2048 >> $out || exit 1
done
cat $out
