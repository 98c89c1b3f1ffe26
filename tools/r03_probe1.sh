#!/bin/bash
# round 3, first GPU call: sanity (GPU tests), k_recon sensitivity builds (variants/*.so, wrong output, timing only), the
# memory-side-cache probe (few streams: a level's frames fit the 256 MB) and the two-stream overlap probe
mkdir -p gpurun_out
{
echo "== gpu tests"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== variants, 64 x 120"; tools/variants.sh run 64 120 4
echo "== few streams (base build)"
for n in 2 4 8 16 32; do echo -n "$n streams: "; timeout 200 python tools/kbench.py $n 120 5 2>&1 | tail -1; done
echo "== two half batches on two streams"; timeout 300 python tools/overlap_probe.py 4 2>&1 | tail -5
} > gpurun_out/r03_probe1.txt 2>&1
tail -40 gpurun_out/r03_probe1.txt
