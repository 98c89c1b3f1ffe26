#!/bin/bash
mkdir -p gpurun_out
{
for v in base noadd nopredarith noscatter noidct noall base; do
echo "== $v"; JSMPEG_HIP_RECON_ORDER=0 JSMPEG_HIP_LIB=$PWD/variants/$v.so timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^recon launches\|^{"
done
} > gpurun_out/r04_run8.txt 2>&1
tail -70 gpurun_out/r04_run8.txt
