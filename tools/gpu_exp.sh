#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
(cd $R && timeout 300 tools/micro/gemm_s3_bench > $O/gemm_s3_micro.txt 2>&1; cut -c1-170 $O/gemm_s3_micro.txt)
(bash $R/tools/s3_clock.sh - _a1 _a2 _a3 -zero -const > $O/gemm_s3_clock.txt 2>&1; cat $O/gemm_s3_clock.txt)
