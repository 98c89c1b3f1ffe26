#!/bin/bash
# Round 5: counters + lane occupancy of the CURRENT slice parse (variants/new.so, variants/stats.so), kbench of every other variant.
tag="${1:-r05c}"
ROOT=$(pwd); out=$ROOT/gpurun_out/$tag; mkdir -p $out
shapes=("cfg2_1080p 64 120" "cfg4_2160p 64 24" "cfg4_2160p 16 24")
kb() { JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/kbench.py $2 $3 5 2>&1 | tail -1 | sed 's/recon per level.*//'; }
for so in variants/*.so; do
  n=$(basename $so .so); [ "$n" = stats ] && continue
  for s in "${shapes[@]}"; do set -- $s; echo -n "variant $n | $1 $2 x $3: " >> $out/variants.txt; JSMPEG_HIP_LIB=$ROOT/$so kb $1 $2 $3 >> $out/variants.txt; done
done
if [ -f variants/stats.so ]; then
  for s in "${shapes[@]}"; do
    set -- $s
    echo "== lane occupancy, $1 $2 x $3" >> $out/parse_stats.txt
    JSMPEG_HIP_LIB=$ROOT/variants/stats.so JSMPEG_KBENCH_CONFIG=$1 timeout 300 python tools/parse_stats.py $2 $3 2>&1 | grep -v amdgpu.ids >> $out/parse_stats.txt
  done
fi
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC"
 "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"
)
for s in "${shapes[@]:0:2}"; do
  set -- $s
  echo "== counters, $1 $2 x $3" >> $out/parse_pmc.txt
  i=0
  for g in "${groups[@]}"; do
    d=$out/pmc_g$i; rm -rf $d
    (cd /tmp && TMPDIR=/tmp JSMPEG_HIP_LIB=$ROOT/variants/new.so JSMPEG_KBENCH_CONFIG=$1 timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -- python $ROOT/tools/kbench.py $2 $3 3 > /dev/null 2> $d.err)
    python tools/pmc_dump.py $d | grep "^k_parse" >> $out/parse_pmc.txt
    rm -rf $d $d.err
    i=$((i+1))
  done
done
cat $out/variants.txt $out/parse_stats.txt $out/parse_pmc.txt
