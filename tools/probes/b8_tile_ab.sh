#!/bin/bash
# GPU box: BASELINE configs[1] (batch 8) under the Winograd tile / operand-form choices:  tools/probes/b8_tile_ab.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-b8tile}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for E in "DT_NOP=0" "DT_H2_MINFRAMES=0 DT_S3_MINROWS=1 DT_WINO_TILE=4" "DT_S3_MINROWS=1 DT_WINO_TILE=4" "DT_H2_MINFRAMES=0 DT_S3_MINROWS=1 DT_WINO_TILE=4 DT_S3_HALF=-1"; do
  echo "== $E"; env $E timeout 300 python tools/b8_profile.py 2>&1 | grep -v "^$" | head -70
done > $O/out.txt 2>&1
grep -E "^==|^wall|sum of" $O/out.txt
