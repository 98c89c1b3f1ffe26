#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== latency probe with the host trace"; JSMPEG_HIP_TRACE=1 timeout 300 python tools/latency_probe.py 2>&1 | grep -v "^$" | head -30
} > gpurun_out/r03_probe3.txt 2>&1
tail -40 gpurun_out/r03_probe3.txt
