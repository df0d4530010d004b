#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/b8_profile.py 2>&1 | grep -E "wall|conv_1[0257]|decode|sum" | tee $O/b8.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "handover or configs1 or detector" 2>&1 | tail -3 | tee -a $O/b8.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -q -x -k "configs1" 2>&1 | tail -3 | tee -a $O/b8.txt
