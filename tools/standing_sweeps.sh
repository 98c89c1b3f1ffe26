#!/bin/bash
# On the GPU box: the standing fuzz sweeps against the oracle with fresh seeds (one line each) -- what profiles/rNN_fuzz.txt records.
#   tools/standing_sweeps.sh [seed offset]
k="${1:-0}"
run() { echo "$*"; "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
run python tools/fuzz_parity.py 1500 $((670007 + k))
run python tools/fuzz_live.py 600 $((43 + k))
run python tools/fuzz_abi_chunks.py 400 $((680008 + k))
run python tools/fuzz_corrupt.py 400 $((85 + k))
echo "FUZZ_ORDERED=2"; FUZZ_ORDERED=2 python tools/fuzz_parity.py 400 $((690009 + k)) 2>&1 | grep -v amdgpu.ids | tail -1
run python tools/fuzz_mp2.py 200 $((7 + k))
run python tools/fuzz_live_audio.py 500 $((9 + k))
