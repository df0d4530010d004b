#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_multi.py -q -x -k "partitions" 2>&1 | tail -4 | tee $O/out.txt
