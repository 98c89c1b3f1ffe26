#!/bin/bash
# Round 5: the dieted slice parse on the box -- GPU tests, then the header threshold and priority sweeps against
# variants/base.so (the round-4 kernel, same box).   tools/r05_parse_sweep.sh <tag>
tag="${1:-r05b}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests.txt
shapes=("cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300" "cfg1_720p 64 120" "cfg1_720p 1 360")
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 5 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for s in "${shapes[@]}"; do
  set -- $s
  [ -f variants/base.so ] && { echo -n "r04 kernel | $1 $2 x $3: "; JSMPEG_HIP_LIB=$ROOT/variants/base.so kb $1 $2 $3; } >> $out/sweep.txt
  echo -n "new, rule   | $1 $2 x $3: " >> $out/sweep.txt; kb $1 $2 $3 >> $out/sweep.txt
  for t in 12 16 20 24 28 32 40; do
    echo -n "new, T_COLD=$t prio=0 | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_T_COLD=$t JSMPEG_HIP_PARSE_PRIO=0 kb $1 $2 $3 >> $out/sweep.txt
  done
  for p in 0 64 256 1024; do
    echo -n "new, rule T_COLD, prio=$p | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_PARSE_PRIO=$p kb $1 $2 $3 >> $out/sweep.txt
  done
done
for so in variants/*.so; do
  n=$(basename $so .so); [ "$n" = base ] && continue; [ "$n" = stats ] && continue
  for s in "${shapes[@]:0:3}"; do set -- $s; echo -n "variant $n | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/$so kb $1 $2 $3 >> $out/sweep.txt; done
done
tail -3 $out/gpu_tests.txt; cat $out/sweep.txt
