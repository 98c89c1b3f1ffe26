#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "== latency"; timeout 300 python tools/latency_probe.py 2>&1 | tail -5
echo "== configs"; bash tools/configs_sweep.sh 2>&1
} > gpurun_out/r03_probe6.txt 2>&1
tail -30 gpurun_out/r03_probe6.txt
