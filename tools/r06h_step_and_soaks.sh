mkdir -p gpurun_out
for n in 12 24 48 120 240 480; do echo "== 64 x $n"; timeout 600 python tools/kbench.py 64 $n 5 2>&1 | grep -v amdgpu.ids | tail -4; done > gpurun_out/r06h_step_vs_pictures.txt 2>&1
timeout 300 python tools/soak_live.py --ticks 60000 --seed 11 --out gpurun_out/r06h_soak_live.txt > /dev/null 2>&1; tail -n 2 gpurun_out/r06h_soak_live.txt
timeout 300 python tools/soak_live_audio.py --ticks 60000 --seed 11 --out gpurun_out/r06h_soak_live_audio.txt > /dev/null 2>&1; tail -n 2 gpurun_out/r06h_soak_live_audio.txt
timeout 400 python tools/soak_ordered.py --passes 2000 --out gpurun_out/r06h_soak_ordered.txt > /dev/null 2>&1; tail -n 1 gpurun_out/r06h_soak_ordered.txt
cat gpurun_out/r06h_step_vs_pictures.txt
