#!/bin/bash
# Dev tool: build the fused kernel with one phase removed at a time (HOMMX_ABLATE_*; results are then WRONG, timing only)
# into tools/bin/, and (on the GPU box) time each:   tools/ablate_fused.sh build   |   tools/ablate_fused.sh run
set -e
cd "$(dirname "$0")/.."
V="${ABLATE_SET:-NONE SWEEP GEMM2 RL WNEXT SNEXT SWEEPLDS}"
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for a in $V; do
    /opt/rocm/bin/hipcc -DHOMMX_ABLATE_$a -DHOMMX_FUSED_WAVES_PER_SIMD=2 -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c hommx_amd/csrc/fused2d.hip -o tools/bin/fused_$a.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_$a.so hommx_amd/csrc/api.o tools/bin/fused_$a.o hommx_amd/csrc/blocked.o hommx_amd/csrc/calibrate.o
  done
else
  for a in $V; do echo -n "$a: "; HOMMX_LIB=$PWD/tools/bin/lib_$a.so python tools/time_only.py; done
fi
