#!/bin/bash
mkdir -p gpurun_out/r06t
timeout -k 10 200 ./ab/hbm_read_probe > gpurun_out/r06t/hbm_read_probe.txt 2>&1; echo rc $?
cat gpurun_out/r06t/hbm_read_probe.txt
