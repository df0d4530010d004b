#!/bin/bash
# timing build of the LDS-staged fused F(4x4) kernel: -DDT_S4_TIMING (+ extra flags in $1) -> tools/_probe_builds/libmi355_dt_s4tt$2.so
set -e
cd "$(dirname "$0")/.."
D=tools/_probe_builds; mkdir -p $D
C=object_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDT_S4_TIMING -fno-slp-vectorize $1 -c $C/wino4s_fused.hip -o $D/wino4s_fused_tt$2.o
OBJS=$(ls $C/*.o | grep -v wino4s_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_s4tt$2.so $D/wino4s_fused_tt$2.o $OBJS
echo built $D/libmi355_dt_s4tt$2.so
