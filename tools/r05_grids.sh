#!/bin/bash
# Late round 5: the slice parse's grid for passes without tickets (JSMPEG_HIP_PARSE_EVEN = 0 packed / 1 the rule / 2 everything spread).
tag="${1:-r05ae}"; ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
for r in 1 2; do
  for c in "cfg4_2160p 64 24" "cfg4_2160p 48 24" "cfg0_240p_intra 40 300" "cfg1_720p 32 120" "cfg4_2160p 32 24" "cfg4_2160p 16 24" "cfg4_2160p 8 24" "cfg1_720p 1 360" "cfg2_1080p 4 120"; do
    set -- $c
    for e in 0 1 2; do
      echo -n "JSMPEG_HIP_PARSE_EVEN=$e (run $r) | $1 $2 x $3: " >> $out/grids.txt
      JSMPEG_HIP_PARSE_EVEN=$e JSMPEG_KBENCH_CONFIG=$1 timeout 200 python tools/kbench.py $2 $3 8 2>&1 | tail -1 | sed 's/recon per level.*//' >> $out/grids.txt
    done
  done
done
for e in 0 1 2; do echo "== JSMPEG_HIP_PARSE_EVEN=$e: tools/latency_probe.py" >> $out/grids.txt; JSMPEG_HIP_PARSE_EVEN=$e timeout 200 python tools/latency_probe.py 2>&1 | grep "one-picture ABI.*\(streaming\|decode-ahead 0\)" >> $out/grids.txt; done
cat $out/grids.txt
