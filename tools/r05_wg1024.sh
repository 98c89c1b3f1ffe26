#!/bin/bash
tag="${1:-r05s}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for so in variants/*.so; do n=$(basename $so .so); echo -n "$n parity: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/$so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ordered.py -x -q 2>&1 | tail -1 >> $out/sweep.txt; done
for rep in 1 2; do for s in "cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300" "cfg1_720p 1 360" "cfg1_720p 64 120"; do
  set -- $s
  for so in variants/*.so; do n=$(basename $so .so); echo -n "$n (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/$so kb $1 $2 $3 >> $out/sweep.txt; done
done; done
cat $out/sweep.txt
