#!/bin/bash
# Round 5, late: the slice parse's dependent LDS round trips (carried bit window, the pair entry's halves in one trip, the
# header step's five looks as one to three).  On the GPU box:   tools/r05_latency.sh <tag> [variant names...]
# GPU suite on the product library, then tools/kbench.py per variants/<name>.so and shape, alternating, REPS rounds.
tag="${1:-r05y}"; shift
names="${@:-base new}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?" >> $out/gpu_tests.txt
fi
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 6 2>&1 | tail -1 | sed 's/recon per level.*//'; }
IFS=';' read -ra shapes <<< "${SHAPES:-cfg2_1080p 64 120;cfg4_2160p 64 24;cfg4_2160p 16 24;cfg1_720p 1 360}"
for rep in $(seq 1 ${REPS:-2}); do for s in "${shapes[@]}"; do
  set -- $s
  for n in $names; do
    echo -n "$n (run $rep) | $1 $2 x $3: " >> $out/sweep.txt; JSMPEG_HIP_LIB=$ROOT/variants/$n.so kb $1 $2 $3 >> $out/sweep.txt
  done
done; done
[ -f $out/gpu_tests.txt ] && tail -2 $out/gpu_tests.txt; cat $out/sweep.txt
