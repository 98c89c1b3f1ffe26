#!/bin/bash
tag="${1:-r05t}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for rep in 1 2; do for s in "cfg4_2160p 64 24" "cfg4_2160p 32 24" "cfg4_2160p 16 24" "cfg2_1080p 16 120" "cfg2_1080p 32 60" "cfg1_720p 1 360" "cfg1_720p 32 120"; do
  set -- $s
  echo -n "second-round workgroups rotated (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; kb $1 $2 $3 >> $out/sweep.txt
  echo -n "before (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/variants/norot.so kb $1 $2 $3 >> $out/sweep.txt
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1 >> $out/sweep.txt
cat $out/sweep.txt
