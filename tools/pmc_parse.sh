#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of k_parse from one rocprofv3 --pmc pass each over tools/kbench.py (the current build).
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/gpurun_out/pmc_$c -- python $ROOT/tools/kbench.py 64 120 3 > /dev/null 2>&1
done
cd $ROOT; python tools/pmc_dump.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 2>/dev/null | grep -E "k_parse|k_recon"
