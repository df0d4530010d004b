#!/bin/bash
set -e
cd "$(dirname "$0")/.."
D=object_tracking_amd/ablate; mkdir -p $D
C=object_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDT_W4_TIMING -c $C/wino4_fused.hip -o $D/wino4_fused_tt.o
OBJS=$(ls $C/*.o | grep -v wino4_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_w4tt.so $D/wino4_fused_tt.o $OBJS
echo built $D/libmi355_dt_w4tt.so
