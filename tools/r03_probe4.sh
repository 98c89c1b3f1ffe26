#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu parity tests (product build)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for v in w2 w8; do echo "== gpu parity tests, $v"; JSMPEG_HIP_LIB=$PWD/variants/$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2; done
echo "== variants, 64 x 120"; tools/variants.sh run 64 120 4
echo "== cfg0 64 x 300"; for v in old base w2; do echo -n "$v: "; JSMPEG_KBENCH_CONFIG=cfg0_240p_intra JSMPEG_HIP_LIB=$PWD/variants/$v.so timeout 200 python tools/kbench.py 64 300 4 2>&1 | tail -1; done
} > gpurun_out/r03_probe4.txt 2>&1
tail -40 gpurun_out/r03_probe4.txt
