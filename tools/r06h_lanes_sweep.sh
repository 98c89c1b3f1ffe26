#!/bin/bash
# On the GPU box: slices per wavefront of the slice parse (JSMPEG_HIP_PARSE_LANES) on mid-size passes of cfg2's content
# (64 streams x 6 / 12 / 24 / 48 pictures): parse_ms per setting -> gpurun_out/r06h_lanes_sweep.txt
for n in 6 12 24 48; do
  for l in 0 8 16 24 32 48; do
    if [ "$l" = 0 ]; then unset JSMPEG_HIP_PARSE_LANES; else export JSMPEG_HIP_PARSE_LANES=$l; fi
    echo -n "64 x $n lanes ${l}: "; JSMPEG_HIP_PARSE_SAY=1 timeout 300 python tools/kbench.py 64 $n 6 2>&1 | grep -v amdgpu.ids | grep "k_parse\|index_ms" | tail -2 | tr '\n' ' ' | cut -c1-330; echo
  done
done
