#!/bin/bash
# On the GPU box, the tree with JM_ORDER_DISTANCE 400: the GPU suite, the shapes the ordered launch serves (kbench), coded video with
# the ordered launch left to itself (JSMPEG_HIP_RECON_DENSE=0) and by the engine's rule, the ordered launch's soak and fuzz
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06n_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r06n_pytest.log | tail -2
k() { echo -n "$1: "; shift; "$@" 2>&1 | grep -v amdgpu.ids | grep "reconstruct:\|index_ms" | tr '\n' ' ' | cut -c1-330; echo; }
k "cfg2 64 x 120" python tools/kbench.py 64 120 8
k "cfg2 64 x 48" python tools/kbench.py 64 48 8
k "cfg2 16 x 120" python tools/kbench.py 16 120 8
k "cfg2 4 x 96" python tools/kbench.py 4 96 8
k "720p 64 x 120" env JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 64 120 8
k "720p 1 x 360" env JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 1 360 8
k "2160p 64 x 24" env JSMPEG_KBENCH_CONFIG=cfg4_2160p python tools/kbench.py 64 24 6
e() { echo -n "$1: "; shift; "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['gpu_phases_ms'], d['reconstruct'], d['frames_per_s'])"; }
e "coded video, the engine's rule" python tools/enc_content_bench.py 64 10 6
e "coded video, JSMPEG_HIP_RECON_DENSE=0 (the ordered launch by itself)" env JSMPEG_HIP_RECON_DENSE=0 python tools/enc_content_bench.py 64 10 6
timeout 400 python tools/soak_ordered.py --passes 2000 --out gpurun_out/r06n_soak_ordered.txt > /dev/null 2>&1; tail -n 8 gpurun_out/r06n_soak_ordered.txt | cut -c1-260
echo "FUZZ_ORDERED=2"; FUZZ_ORDERED=2 python tools/fuzz_parity.py 600 697009 2>&1 | grep -v amdgpu.ids | tail -1
python tools/fuzz_parity.py 1000 698010 2>&1 | grep -v amdgpu.ids | tail -1
