#!/bin/bash
# Round 5: the header threshold of intra-only wavefronts separately from the pass's
tag="${1:-r05l}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for tp in 24 28 32; do for ti in 0 4 8 12 16; do
  echo -n "T_COLD=$tp intra=$ti | cfg2_1080p 64 x 120: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD=$tp JSMPEG_HIP_T_COLD_INTRA=$ti kb cfg2_1080p 64 120 >> $out/sweep.txt
done; done
for ti in 0 8 12; do
  echo -n "rule, intra=$ti | cfg1_720p 64 x 120: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD_INTRA=$ti kb cfg1_720p 64 120 >> $out/sweep.txt
  echo -n "rule, intra=$ti | cfg1_720p 1 x 360: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD_INTRA=$ti kb cfg1_720p 1 360 >> $out/sweep.txt
  echo -n "rule, intra=$ti | cfg2_1080p 4 x 120: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD_INTRA=$ti kb cfg2_1080p 4 120 >> $out/sweep.txt
done
for tl in 0 1; do for s in "cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300" "cfg1_720p 64 120"; do
  set -- $s
  echo -n "rule, live-scaled threshold=$tl | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD_LIVE=$tl kb $1 $2 $3 >> $out/sweep.txt
done; done
cat $out/sweep.txt
