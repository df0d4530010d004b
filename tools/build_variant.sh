#!/bin/bash
# A/B build of the library with one translation unit recompiled under extra flags (timing experiments, selected with MI355_DT_LIB):
#   tools/build_variant.sh <name> <unit.hip> <flags...>   ->   tools/_probe_builds/libmi355_dt_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; UNIT=$2; shift 2
C=object_tracking_amd/csrc; D=tools/_probe_builds; mkdir -p $D
EXTRA=""; [ "$UNIT" = "wino4s_fused.hip" ] && EXTRA="-fno-slp-vectorize"
case "$UNIT" in decode.hip|targets.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-gpu-rdc -Wno-unused-function $EXTRA "$@" -c $C/$UNIT -o $D/${UNIT%.hip}_$NAME.o
OBJS=$(ls $C/*.o | grep -v "/${UNIT%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libmi355_dt_$NAME.so $D/${UNIT%.hip}_$NAME.o $OBJS
echo $D/libmi355_dt_$NAME.so
