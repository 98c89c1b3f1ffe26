#!/bin/bash
# On the GPU box: the reconstruct level by level (JSMPEG_HIP_RECON_ORDER=0) against the engine's own choice, shape by shape (kbench)
k() { for o in auto 0; do if [ $o = auto ]; then unset JSMPEG_HIP_RECON_ORDER; else export JSMPEG_HIP_RECON_ORDER=$o; fi; echo -n "$1 order $o: "; "${@:2}" 2>&1 | grep -v amdgpu.ids | grep "reconstruct:\|index_ms" | tr '\n' ' ' | sed 's/recon per level.*|//' | cut -c1-260; echo; done; }
k "1080p 64 x 120" python tools/kbench.py 64 120 8
k "1080p 64 x 48" python tools/kbench.py 64 48 8
k "1080p 64 x 24" python tools/kbench.py 64 24 8
k "1080p 64 x 12" python tools/kbench.py 64 12 8
k "1080p 32 x 120" python tools/kbench.py 32 120 8
k "1080p 16 x 120" python tools/kbench.py 16 120 8
k "1080p 8 x 120" python tools/kbench.py 8 120 8
k "1080p 4 x 96" python tools/kbench.py 4 96 8
k "720p 64 x 120" env JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 64 120 8
k "720p 16 x 120" env JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 16 120 8
k "720p 1 x 360" env JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 1 360 8
