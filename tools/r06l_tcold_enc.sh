#!/bin/bash
# On the GPU box: the header step's queue threshold (JSMPEG_HIP_T_COLD, whole pass) and the ring's service form on the
# encoder-made content at 16 Mbit/s -> gpurun_out/r06l_tcold_enc.txt (parse_ms per setting)
for split in 0 1; do for t in 0 2 4 8 12 16 32 48; do
  export JSMPEG_HIP_PARSE_SPLIT=$split
  if [ $t = 0 ]; then unset JSMPEG_HIP_T_COLD; else export JSMPEG_HIP_T_COLD=$t; fi
  echo -n "split $split t_cold $t: "; python tools/enc_content_bench.py 64 10 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['gpu_phases_ms']['parse_ms'], d['gpu_phases_ms']['total_ms'], d['frames_per_s'])"
done; done
