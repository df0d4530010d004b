#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in c1pf1 c1pf2 c1pf2sync c1pf1 c1pf2; do
  MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_$v.so timeout 300 python tools/c1_time.py 2>&1 | tail -1 | tee -a $O/c1.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "conv1 or detector or first_layer" 2>&1 | tail -3 | tee -a $O/c1.txt
