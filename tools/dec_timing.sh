#!/bin/bash
# debug build of the library with phase timestamps inside decode_nms_kernel (tools/dec_timing.py reads them)
set -e
cd "$(dirname "$0")/.."
D=tools/_probe_builds; mkdir -p $D
C=object_tracking_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -DDT_DEC_TIMING -c $C/decode.hip -o $D/decode_tt.o
OBJS=$(ls $C/*.o | grep -v "/decode.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_dectt.so $D/decode_tt.o $OBJS
echo built $D/libmi355_dt_dectt.so
