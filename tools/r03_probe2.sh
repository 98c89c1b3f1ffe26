#!/bin/bash
# round 3, second GPU call: the pair-table slice parse -- GPU tests with the product build, then the schedule variants
mkdir -p gpurun_out
{
echo "== gpu tests (product build = pair table)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== gpu parity tests, dc2 variant"; JSMPEG_HIP_LIB=$PWD/variants/pair_dc2.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
echo "== variants, 64 x 120"; tools/variants.sh run 64 120 4
} > gpurun_out/r03_probe2.txt 2>&1
tail -30 gpurun_out/r03_probe2.txt
