#!/bin/bash
# Tuning experiments on the GPU box: rebuild libjsmpeg_hip.so with -D overrides, run a short bench, print the phase times.
#   tools/variant_bench.sh "name1:-DJM_X=1 -DJM_Y=2" "name2:..."
mkdir -p gpurun_out
for v in "$@"; do
  name="${v%%:*}"; defs="${v#*:}"
  JSMPEG_HIP_DEFS="$defs" JSMPEG_HIP_FORCE=1 python -m jsmpeg_amd.build hip > gpurun_out/build_$name.log 2>&1 || { echo "$name: BUILD FAILED"; tail -5 gpurun_out/build_$name.log; continue; }
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err || { echo "$name: BENCH FAILED"; tail -3 gpurun_out/v_$name.err; continue; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open("gpurun_out/v_%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["roofline"]["phases_ms"])
PY
done
