#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (ordered)"; timeout 900 python -m pytest tests/test_gpu_ordered.py -q 2>&1 | tail -15
echo "== base"; JSMPEG_HIP_LIB=$PWD/variants/base.so JSMPEG_KBENCH_ORDERS=0,1,2,3,4,0,2 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct"
} > gpurun_out/r04_run4.txt 2>&1
tail -60 gpurun_out/r04_run4.txt
