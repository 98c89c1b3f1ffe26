#!/bin/bash
# On the GPU box: encoder-made content (a) with two batches in flight, (b) through timing variants of the parse with 3 / 4
# coefficient steps per turn (variants/*.so: tools/variants.sh build base: rep3:"-DJM_COEF_REPEAT=3" rep4:"-DJM_COEF_REPEAT=4")
mkdir -p gpurun_out
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('mbit_per_s_per_stream_at_30fps','frames_per_s','gpu_phases_ms')}); t=d.get('two_batches_in_flight'); print('   two batches in flight:', {k:t[k] for k in ('value','ms_per_pass','parity')}) if t else None"; }
echo "== two batches in flight, GOPs 0,2,4,6"; python tools/enc_content_bench.py 64 10 6 --two --out gpurun_out/r06k_enc_content_16mbit.json 2> gpurun_out/r06k_last.err | show || tail -3 gpurun_out/r06k_last.err
echo "== two batches in flight, all GOPs"; python tools/enc_content_bench.py 64 10 6 --gops all --two --out gpurun_out/r06k_enc_content_all_gops.json 2> gpurun_out/r06k_last.err | show || tail -3 gpurun_out/r06k_last.err
for so in variants/base.so variants/rep3.so variants/rep4.so; do
  export JSMPEG_HIP_LIB=$PWD/$so
  echo "== $(basename $so .so): encoder content GOPs 0,2,4,6"; python tools/enc_content_bench.py 64 10 5 2> gpurun_out/r06k_last.err | show || tail -3 gpurun_out/r06k_last.err
  echo "== $(basename $so .so): encoder content all GOPs"; python tools/enc_content_bench.py 64 10 5 --gops all 2> gpurun_out/r06k_last.err | show || tail -3 gpurun_out/r06k_last.err
  echo -n "== $(basename $so .so): cfg2 64 x 120: "; python tools/kbench.py 64 120 5 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
  echo -n "== $(basename $so .so): cfg4 64 x 24: "; JSMPEG_KBENCH_CONFIG=cfg4_2160p python tools/kbench.py 64 24 5 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120
done
