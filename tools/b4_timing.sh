#!/bin/bash
# timing / ablation builds of wino4b_fused.hip: tools/b4_timing.sh "<extra flags>" <suffix>  ->  tools/_probe_builds/libmi355_dt_b4<suffix>.so
set -e
cd "$(dirname "$0")/.."
D=tools/_probe_builds; mkdir -p $D
C=object_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-gpu-rdc -Wno-unused-function $1 -c $C/wino4b_fused.hip -o $D/wino4b_fused_$2.o
OBJS=$(ls $C/*.o | grep -v wino4b_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_b4$2.so $D/wino4b_fused_$2.o $OBJS
echo built $D/libmi355_dt_b4$2.so
