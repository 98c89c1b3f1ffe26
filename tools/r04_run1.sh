#!/bin/bash
# round 4, run 1: ordered launch parity + sweep
mkdir -p gpurun_out
{
echo "== gpu tests (ordered + parity)"; timeout 900 python -m pytest tests/test_gpu_ordered.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15
echo "== sweep, nt stores"; JSMPEG_HIP_LIB=$PWD/variants/nt.so JSMPEG_KBENCH_ORDERS=0,1,2,3,4,8 timeout 600 python tools/kbench.py 64 120 5 2>&1 | grep -v "^recon launches" 
echo "== sweep, plain stores"; JSMPEG_HIP_LIB=$PWD/variants/plain.so JSMPEG_KBENCH_ORDERS=0,2,3,4 timeout 600 python tools/kbench.py 64 120 5 2>&1 | grep -v "^recon launches"
} > gpurun_out/r04_run1.txt 2>&1
tail -60 gpurun_out/r04_run1.txt
