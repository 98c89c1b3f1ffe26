#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (ordered)"; timeout 900 python -m pytest tests/test_gpu_ordered.py -q 2>&1 | tail -15
for v in base noepi nopoll noboth; do
echo "== $v"; JSMPEG_HIP_LIB=$PWD/variants/$v.so JSMPEG_KBENCH_ORDERS=0,2,4 timeout 600 python tools/kbench.py 64 120 5 2>&1 | grep "^order\|^reconstruct"
done
} > gpurun_out/r04_run2.txt 2>&1
tail -60 gpurun_out/r04_run2.txt
