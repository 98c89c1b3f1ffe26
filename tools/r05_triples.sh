#!/bin/bash
tag="${1:-r05r}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests.txt
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for rep in 1 2 3; do for s in "cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300" "cfg1_720p 1 360"; do
  set -- $s
  echo -n "two symbols + end_of_block per look (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; kb $1 $2 $3 >> $out/sweep.txt
  echo -n "before (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/variants/notriples.so kb $1 $2 $3 >> $out/sweep.txt
done; done
tail -2 $out/gpu_tests.txt; cat $out/sweep.txt
