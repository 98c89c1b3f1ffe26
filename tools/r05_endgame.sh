#!/bin/bash
# Round 5: the end game of a ticketed parse -- the last K batches drawn by two wavefronts per SIMD only
tag="${1:-r05q}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for rep in 1 2; do
for s in "cfg2_1080p 64 120" "cfg1_720p 64 120" "cfg0_240p_intra 64 300"; do
  set -- $s
  for k in 0 256 512 1024 1536 2048 3072 4096; do
    echo -n "endgame=$k (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_ENDGAME=$k kb $1 $2 $3 >> $out/sweep.txt
  done
done
done
JSMPEG_HIP_PARSE_ENDGAME=1024 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1 >> $out/sweep.txt
cat $out/sweep.txt
