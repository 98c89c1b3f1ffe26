#!/bin/bash
tag="${1:-r05w}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for s in "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg2_1080p 16 120" "cfg1_720p 1 360"; do
  set -- $s
  for y in 0 1 2 4 8; do echo -n "short batches sleep $y per turn | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_YIELD=$y kb $1 $2 $3 >> $out/sweep.txt; done
done
cat $out/sweep.txt
