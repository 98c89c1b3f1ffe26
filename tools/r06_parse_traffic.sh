#!/bin/bash
# round 6 on the box (the round-5 review's item 6): what the slice parse fetches per pass (FETCH_SIZE: 8.2 x the compressed
# bytes -- a lane's 16-byte refills come back to their 128-byte line up to eight times) against the ring service's rule
# JM_REFILL_MIN (slice_parse.h): timing over five shapes (kbench, alternating processes) and one rocprofv3 --pmc pass per
# counter and variant on cfg2.     here: tools/variants.sh build min1:"-DJM_REFILL_MIN=1" min2:"-DJM_REFILL_MIN=2" min3:"-DJM_REFILL_MIN=3"
ROOT=$(pwd); export TMPDIR=/tmp
shape() { for round in 1 2; do for so in variants/*.so; do echo -n "$(basename $so .so) $1 | "; env JSMPEG_KBENCH_CONFIG=$2 JSMPEG_HIP_LIB=$ROOT/$so timeout 300 python tools/kbench.py $3 $4 6 2>&1 | grep -v amdgpu.ids | tail -1; done; done; }
shape "cfg2 64 x 120" cfg2_1080p 64 120
shape "cfg4 64 x 24" cfg4_2160p 64 24
shape "cfg4 16 x 24" cfg4_2160p 16 24
shape "cfg0 64 x 300" cfg0_240p_intra 64 300
shape "cfg1 1 x 360" cfg1_720p 1 360
shape "cfg1 64 x 120" cfg1_720p 64 120
for so in variants/*.so; do
  v=$(basename $so .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${v}_$c
    (cd /tmp && JSMPEG_HIP_LIB=$ROOT/$so timeout 300 rocprofv3 --kernel-trace --pmc $c -d $ROOT/gpurun_out/pmc_${v}_$c -- python $ROOT/tools/kbench.py 64 120 3 > /dev/null 2>&1)
  done
  echo "== $v: counters on cfg2 (per-dispatch averages; FETCH_SIZE in KiB, x 2 on gfx950 for streamed reads; WRITE_SIZE in KiB)"
  python tools/pmc_dump.py gpurun_out/pmc_${v}_FETCH_SIZE gpurun_out/pmc_${v}_WRITE_SIZE 2>/dev/null | grep -E "k_parse"
done
