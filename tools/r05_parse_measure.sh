#!/bin/bash
# Round 5, item 1(a): what the slice parse's lanes are doing.  On the GPU box:
#   tools/r05_parse_measure.sh <tag>
# 1. per-step-kind lane occupancy of k_parse (variants/stats.so = a -DJM_PARSE_STATS build, tools/parse_stats.py)
#    for cfg2 64 x 120, cfg4 64 x 24, cfg4 16 x 24, cfg0 64 x 300;
# 2. the divergence counters of k_parse (one rocprofv3 --pmc pass per group, counters only + kernel trace) for cfg2 and cfg4;
# 3. kbench timings of every variants/*.so that is not the stats build, on the three shapes.
tag="${1:-r05a}"
ROOT=$(pwd)
out=$ROOT/gpurun_out/$tag
mkdir -p $out
shapes=("cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24" "cfg0_240p_intra 64 300")

if [ -f variants/stats.so ]; then
  for s in "${shapes[@]}"; do
    set -- $s
    echo "== lane occupancy, $1 $2 x $3" >> $out/parse_stats.txt
    JSMPEG_HIP_LIB=$ROOT/variants/stats.so JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/parse_stats.py $2 $3 >> $out/parse_stats.txt 2>&1
  done
fi

groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE"
)
for s in "${shapes[@]:0:2}"; do
  set -- $s
  echo "== counters, $1 $2 x $3 (per-kernel averages over the passes of tools/kbench.py)" >> $out/parse_pmc.txt
  i=0
  for g in "${groups[@]}"; do
    d=$out/pmc_g$i; rm -rf $d
    (cd /tmp && TMPDIR=/tmp JSMPEG_KBENCH_CONFIG=$1 timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -- python $ROOT/tools/kbench.py $2 $3 3 > /dev/null 2> $d.err)
    python tools/pmc_dump.py $d | grep "^k_parse\|^k_recon " >> $out/parse_pmc.txt
    rm -rf $d $d.err
    i=$((i+1))
  done
done

for so in variants/*.so; do
  n=$(basename $so .so)
  [ "$n" = stats ] && continue
  for s in "${shapes[@]}"; do
    set -- $s
    echo -n "$n | $1 $2 x $3: " >> $out/variants.txt
    JSMPEG_HIP_LIB=$ROOT/$so JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 5 2>&1 | tail -1 | cut -c1-130 >> $out/variants.txt
  done
done
cat $out/parse_stats.txt $out/parse_pmc.txt $out/variants.txt
