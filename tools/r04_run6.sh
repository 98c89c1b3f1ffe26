#!/bin/bash
mkdir -p gpurun_out
{
echo "== gpu tests (all)"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8
echo "== cfg2"; timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg0"; JSMPEG_KBENCH_CONFIG=cfg0_240p_intra timeout 600 python tools/kbench.py 64 300 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg1 64"; JSMPEG_KBENCH_CONFIG=cfg1_720p timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg1 64, levels"; JSMPEG_HIP_RECON_ORDER=0 JSMPEG_KBENCH_CONFIG=cfg1_720p timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg4 64"; JSMPEG_KBENCH_CONFIG=cfg4_2160p timeout 600 python tools/kbench.py 64 24 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg4 64, levels"; JSMPEG_HIP_RECON_ORDER=0 JSMPEG_KBENCH_CONFIG=cfg4_2160p timeout 600 python tools/kbench.py 64 24 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg2 coherent pan (mv_jitter 2)"; JSMPEG_SYNTH_MV_JITTER=2 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
echo "== cfg2 coherent pan, levels"; JSMPEG_HIP_RECON_ORDER=0 JSMPEG_SYNTH_MV_JITTER=2 timeout 600 python tools/kbench.py 64 120 6 2>&1 | grep "^order\|^reconstruct\|^{"
} > gpurun_out/r04_run6.txt 2>&1
tail -70 gpurun_out/r04_run6.txt
