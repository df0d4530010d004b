#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-b4u}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export DT_F4B=1
for v in ${VARS:-ttu0 ttu1 ttu2}; do echo "== variant $v"; MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_b4$v.so timeout 300 python tools/b4_timing.py conv_2 ${FRAMES:-1440} 2>&1 | grep -v amdgpu.ids | head -8; done | tee $O/upath.txt
