#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (ordered)"; timeout 1200 python -m pytest tests/test_gpu_ordered.py -q -x 2>&1 | tail -4
echo "== cfg2"; timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg1 64"; JSMPEG_KBENCH_CONFIG=cfg1_720p timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg4 64"; JSMPEG_KBENCH_CONFIG=cfg4_2160p timeout 600 python tools/kbench.py 64 24 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg4 16"; JSMPEG_KBENCH_CONFIG=cfg4_2160p timeout 600 python tools/kbench.py 16 24 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg4 16 levels"; JSMPEG_HIP_RECON_ORDER=0 JSMPEG_KBENCH_CONFIG=cfg4_2160p timeout 600 python tools/kbench.py 16 24 6 2>&1 | grep "^order\|^reconstruct\|^{"
} > gpurun_out/r04_run7.txt 2>&1
tail -70 gpurun_out/r04_run7.txt
