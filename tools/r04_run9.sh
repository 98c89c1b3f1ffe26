#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (shards + ordered)"; timeout 1200 python -m pytest tests/test_gpu_shards.py tests/test_gpu_ordered.py -q -x 2>&1 | tail -25
} > gpurun_out/r04_run9.txt 2>&1
tail -70 gpurun_out/r04_run9.txt
