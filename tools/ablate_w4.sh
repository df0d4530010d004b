#!/bin/bash
# Timing-only ablation builds of wino4_fused.hip (DT_W4_ABLATE bit mask, see the file).  Numerically wrong by design.
set -e
cd "$(dirname "$0")/.."
D=object_tracking_amd/ablate; mkdir -p $D
C=object_tracking_amd/csrc
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDT_W4_ABLATE=$m -c $C/wino4_fused.hip -o $D/wino4_fused_$m.o
  OBJS=$(ls $C/*.o | grep -v wino4_fused.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_w4abl$m.so $D/wino4_fused_$m.o $OBJS
  echo built $D/libmi355_dt_w4abl$m.so
done
