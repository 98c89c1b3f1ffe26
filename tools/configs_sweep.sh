#!/bin/bash
# The other configurations of SURVEY.md 8d as timings of one pass of the resident batch (tools/kbench.py, 6 repetitions).
for c in "cfg1_720p 64 120" "cfg1_720p 1 360" "cfg2_1080p 64 120" "cfg4_2160p 16 24" "cfg4_2160p 64 24" "cfg0_240p_intra 64 300"; do
  set -- $c
  echo "$1 $2 x $3: $(JSMPEG_KBENCH_CONFIG=$1 timeout 200 python tools/kbench.py $2 $3 6 2>&1 | tail -1)"
done
