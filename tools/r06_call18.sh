#!/bin/bash
set -o pipefail
O=gpurun_out/r06r
mkdir -p $O
export PYTHONPATH=$PWD
timeout -k 10 500 python -m pytest tests/test_gpu_fem.py -q -x -m gpu > $O/fem.txt 2>&1; echo "rc $?" | tee $O/steps.txt
tail -5 $O/fem.txt | cut -c1-300
