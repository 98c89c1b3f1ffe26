#!/bin/bash
# Round 5: intra wavefronts with the DC fused into the block-opening steps + the transposed first-batch mapping: GPU tests, then timings
tag="${1:-r05e}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests.txt
shapes=("cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300" "cfg1_720p 64 120" "cfg1_720p 1 360")
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 5 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for s in "${shapes[@]}"; do
  set -- $s
  echo -n "fused + transposed | $1 $2 x $3: " >> $out/sweep.txt; kb $1 $2 $3 >> $out/sweep.txt
  echo -n "unfused, transposed | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_NOFUSE=1 kb $1 $2 $3 >> $out/sweep.txt
  echo -n "fused, not transposed | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/variants/notranspose.so kb $1 $2 $3 >> $out/sweep.txt
  echo -n "unfused, not transposed | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_NOFUSE=1 JSMPEG_HIP_LIB=$ROOT/variants/notranspose.so kb $1 $2 $3 >> $out/sweep.txt
done
tail -3 $out/gpu_tests.txt; cat $out/sweep.txt
