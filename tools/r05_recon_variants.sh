#!/bin/bash
# Round 5: k_recon timing builds (variants/*.so) on cfg2, ordered launch and per-level launches, same box.
tag="${1:-r05f}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
for so in variants/*.so; do
  n=$(basename $so .so)
  echo "== $n (run $rep), ordered launch" >> $out/recon.txt
  JSMPEG_HIP_LIB=$ROOT/$so timeout 300 python tools/kbench.py 64 120 6 2>&1 | tail -3 | cut -c1-260 >> $out/recon.txt
  if [ $rep = 1 ]; then
    echo "== $n, per-level launches" >> $out/recon.txt
    JSMPEG_HIP_RECON_ORDER=0 JSMPEG_HIP_LIB=$ROOT/$so timeout 300 python tools/kbench.py 64 120 6 2>&1 | tail -3 | cut -c1-260 >> $out/recon.txt
  fi
done
done
cat $out/recon.txt
