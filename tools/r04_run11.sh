#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (all)"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -25
echo "== latency probe"; timeout 600 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r04_run11.txt 2>&1
tail -70 gpurun_out/r04_run11.txt
