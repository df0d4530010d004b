#!/bin/bash
# Timing-only ablation builds of conv_igemm (DT_ABLATE bit mask: 1 no barrier, 2 no global
# loads, 4 no LDS stores, 8 no fragment reads).  Results are numerically wrong by design;
# the variants are selected with MI355_DT_LIB and only ever timed (tools/ablate_run.sh).
set -e
cd "$(dirname "$0")/../.."
D=tools/_probe_builds; mkdir -p $D
C=object_tracking_amd/csrc
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDT_ABLATE=$m -c $C/conv_igemm.hip -o $D/conv_igemm_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_abl$m.so $D/conv_igemm_$m.o $C/conv1.o $C/winograd.o $C/wino_fused.o $C/ingest.o $C/decode.o $C/targets.o $C/recurrent.o $C/network.o
  echo built $D/libmi355_dt_abl$m.so
done
