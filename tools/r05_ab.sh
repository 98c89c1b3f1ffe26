#!/bin/bash
# A/B of parse variants on one shape with long runs: tools/r05_ab.sh <tag> <config> <streams> <frames> <kbench reps> <rounds> names...
tag=$1; cfg=$2; st=$3; fr=$4; kr=$5; rounds=$6; shift 6
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
for rep in $(seq 1 $rounds); do for n in "$@"; do
  echo -n "$n (run $rep) | $cfg $st x $fr: " >> $out/ab.txt
  JSMPEG_HIP_LIB=$ROOT/variants/$n.so JSMPEG_KBENCH_CONFIG=$cfg timeout 300 python tools/kbench.py $st $fr $kr 2>&1 | tail -1 | sed 's/recon per level.*//' >> $out/ab.txt
done; done
cat $out/ab.txt
