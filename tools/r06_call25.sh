#!/bin/bash
O=gpurun_out/r06z
mkdir -p $O
export PYTHONPATH=$PWD
for m in 0 1 2 0; do
  if [ $m = 0 ]; then unset FB_P2_SLOTS; else export FB_P2_SLOTS=$m; fi
  timeout -k 10 200 bash tools/quick_headline.sh > $O/headline_slots$m.txt 2>&1
  echo "slots $m: $(tail -n 1 $O/headline_slots$m.txt | cut -c1-220)"
done
