#!/bin/bash
# round 5, first A/B: MFMA horizontal passes of the DoG, XCD-contiguous fine blocks
mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
timeout -k 10 300 python -m pytest tests/test_gpu_ncc.py -x -q -k "dog or blocks or fine or pfa or xcorr" > $O/t_dog.txt 2>&1; echo "pytest rc $?" | tee -a $O/t_dog.txt
tail -3 $O/t_dog.txt
echo "== dog MF" ; timeout -k 10 200 python tools/microbench_dog.py 2>&1 | tee $O/dog_mf.txt
echo "== dog VALU" ; FEABAS_HIP_DOG_VALU=1 timeout -k 10 200 python tools/microbench_dog.py 2>&1 | tee $O/dog_valu.txt
echo "== fine xcd" ; timeout -k 10 120 python tools/microbench_fine.py 2>&1 | tee $O/fine_xcd.txt
echo "== fine plain" ; FEABAS_HIP_PFA_XCD=0 timeout -k 10 120 python tools/microbench_fine.py 2>&1 | tee $O/fine_plain.txt
echo "== headline new"; timeout -k 10 300 bash tools/quick_headline.sh 2>&1 | tee $O/head_new.txt
echo "== headline old"; FEABAS_HIP_DOG_VALU=1 FEABAS_HIP_PFA_XCD=0 timeout -k 10 300 bash tools/quick_headline.sh 2>&1 | tee $O/head_old.txt
