#!/bin/bash
# Round 5: parity-preserving k_recon variants (cache hints, unaligned prediction loads): GPU parity tests per variant, then timings
tag="${1:-r05j}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
for so in variants/*.so; do
  n=$(basename $so .so)
  echo "== $n: parity (tests/test_gpu_parity.py)" >> $out/recon.txt
  JSMPEG_HIP_LIB=$ROOT/$so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1 >> $out/recon.txt
done
for rep in 1 2; do
for so in variants/*.so; do
  n=$(basename $so .so)
  echo "== $n (run $rep), ordered launch" >> $out/recon.txt
  JSMPEG_HIP_LIB=$ROOT/$so timeout 300 python tools/kbench.py 64 120 6 2>&1 | tail -1 | cut -c1-200 >> $out/recon.txt
done
done
cat $out/recon.txt
