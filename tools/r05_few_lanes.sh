#!/bin/bash
# Round 5: more coefficient steps per turn for wavefronts with few (long) slices; the pair lookup in one LDS round trip.
tag="${1:-r05i}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 5 2>&1 | tail -1 | sed 's/recon per level.*//'; }
shapes=("cfg4_2160p 16 24" "cfg4_2160p 64 24" "cfg1_720p 1 360" "cfg2_1080p 64 120" "cfg2_1080p 4 120" "cfg0_240p_intra 64 300" "cfg4_2160p 4 24")
for s in "${shapes[@]}"; do
  set -- $s
  for fl in 0 4 16; do for fr in 3 4; do
    [ $fl = 0 ] && [ $fr = 4 ] && continue
    echo -n "few_lanes=$fl reps=$fr | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_FEW_LANES=$fl JSMPEG_HIP_PARSE_FEW_REPS=$fr kb $1 $2 $3 >> $out/sweep.txt
  done; done
done
# cfg4 64 x 24 with its intra slices forced into a head of 16 / 8 per wavefront (JSMPEG_HIP_PARSE_HEAD=a,l0,h,l1)
for l in 8 16 32; do for fr in 2 4; do
  echo -n "head 17280 at $l per wavefront, few_lanes=$l reps=$fr | cfg4_2160p 64 x 24: " >> $out/sweep.txt
  JSMPEG_HIP_PARSE_HEAD=17280,$l,17280,$l JSMPEG_HIP_PARSE_FEW_LANES=$l JSMPEG_HIP_PARSE_FEW_REPS=$fr kb cfg4_2160p 64 24 >> $out/sweep.txt
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ordered.py -x -q > $out/gpu_tests.txt 2>&1; echo "rc=$?" >> $out/gpu_tests.txt
JSMPEG_HIP_PARSE_FEW_LANES=64 JSMPEG_HIP_PARSE_FEW_REPS=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $out/gpu_tests_reps4.txt 2>&1; echo "rc=$?" >> $out/gpu_tests_reps4.txt
tail -2 $out/gpu_tests.txt $out/gpu_tests_reps4.txt; cat $out/sweep.txt
