#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (ordered)"; timeout 900 python -m pytest tests/test_gpu_ordered.py -q 2>&1 | tail -15
echo "== base"; JSMPEG_HIP_LIB=$PWD/variants/base.so JSMPEG_KBENCH_ORDERS=0,1,2,3,4 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct"
echo "== vwave"; JSMPEG_HIP_LIB=$PWD/variants/vwave.so JSMPEG_KBENCH_ORDERS=0,2 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct"
echo "== nopoll"; JSMPEG_HIP_LIB=$PWD/variants/nopoll.so JSMPEG_KBENCH_ORDERS=1,2,8 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct"
echo "== nopoll_plain"; JSMPEG_HIP_LIB=$PWD/variants/nopoll_plain.so JSMPEG_KBENCH_ORDERS=0,1,2,8 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct"
} > gpurun_out/r04_run3.txt 2>&1
tail -60 gpurun_out/r04_run3.txt
