#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
CLIPS=96 timeout 900 python tools/dual_partition.py 2>&1 | grep -v "^Native\|amdgpu.ids" | tee $O/out.txt
