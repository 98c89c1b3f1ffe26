#!/bin/bash
# Like variant_bench.sh but with tools/kbench.py (no parity gate): for experiment builds with wrong output.
mkdir -p gpurun_out
for v in "$@"; do
  name="${v%%:*}"; defs="${v#*:}"
  JSMPEG_HIP_DEFS="$defs" JSMPEG_HIP_FORCE=1 python -m jsmpeg_amd.build hip > gpurun_out/build_$name.log 2>&1 || { echo "$name: BUILD FAILED"; tail -5 gpurun_out/build_$name.log; continue; }
  echo -n "$name: "; timeout 300 python tools/kbench.py 2>&1 | tail -1
done
