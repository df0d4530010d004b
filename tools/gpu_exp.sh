#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in "" _o1 "" _o1; do echo "== gemm_s3_bench$v"; timeout 300 tools/micro/gemm_s3_bench$v 2>&1 | cut -c1-170 | tail -5; done
