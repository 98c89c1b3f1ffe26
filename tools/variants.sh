#!/bin/bash
# Tuning experiments without compiling on the GPU box.
#   here      : tools/variants.sh build name:"-DJM_X=1 -DJM_Y" other:"..."   -> variants/<name>.so (git-ignored, travel with gpurun)
#   on the box: tools/variants.sh run [kbench args]                           -> tools/kbench.py once per variants/*.so
mkdir -p variants gpurun_out
if [ "$1" = build ]; then
  shift
  for v in "$@"; do
    name="${v%%:*}"; defs="${v#*:}"; [ "$defs" = "$v" ] && defs=""
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value -I include -I jsmpeg_amd/csrc \
      -o variants/$name.so $defs jsmpeg_amd/csrc/*.hip > variants/build_$name.log 2>&1 && echo "built variants/$name.so [$defs]" || { echo "$name: BUILD FAILED"; tail -5 variants/build_$name.log; }
  done
else
  shift
  for so in variants/*.so; do
    echo -n "$(basename $so .so): "; JSMPEG_HIP_LIB=$PWD/$so timeout 200 python tools/kbench.py "$@" 2>&1 | tail -2 | tr '\n' ' '; echo
  done
fi
