#!/bin/bash
# Like variant_kbench.sh but under rocprofv3 --pmc: prints the per-kernel counter averages of each -D variant.
#   tools/variant_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU" "name1:-DJM_X=1" "name2:..."
mkdir -p gpurun_out
counters="$1"; shift
ROOT=$(pwd)
for v in "$@"; do
  name="${v%%:*}"; defs="${v#*:}"
  JSMPEG_HIP_DEFS="$defs" JSMPEG_HIP_FORCE=1 python -m jsmpeg_amd.build hip > gpurun_out/build_$name.log 2>&1 || { echo "$name: BUILD FAILED"; tail -5 gpurun_out/build_$name.log; continue; }
  rm -rf gpurun_out/pmcv_$name
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc $counters -d $ROOT/gpurun_out/pmcv_$name -- python $ROOT/tools/kbench.py 64 120 2 > /dev/null 2> $ROOT/gpurun_out/pmcv_$name.err)
  echo "== $name"; python tools/pmc_dump.py gpurun_out/pmcv_$name | grep "^k_recon\|^k_parse"
done
