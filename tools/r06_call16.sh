#!/bin/bash
set -o pipefail
O=gpurun_out/r06p
mkdir -p $O
export PYTHONPATH=$PWD
timeout -k 10 300 python tools/prof_ragged.py > $O/prof_ragged.txt 2>&1; echo rc $?
tail -5 $O/prof_ragged.txt | cut -c1-600
