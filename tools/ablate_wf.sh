#!/bin/bash
# timing-only ablation builds of the fused conv_2 kernel (results are wrong by construction): which resource binds it?
set -e
cd "$(dirname "$0")/.."
D=object_tracking_amd/ablate; mkdir -p $D
C=object_tracking_amd/csrc
OBJS=$(ls $C/*.o | grep -v "/wino_fused.o")
for m in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DWF_ABLATE=$m -c $C/wino_fused.hip -o $D/wf_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_wf$m.so $D/wf_$m.o $OBJS
done
echo built
