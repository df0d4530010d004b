#!/bin/bash
# GPU box: wino4s_fused.hip (fp32 fused F(4x4) kernel) parity + timing; DT_F4_MOSAIC=1 (none) against the default (auto)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-f4s}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or conv2_shape or non_square" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
grep -E "^E  |^FAILED|Error" $O/pytest.txt | cut -c1-300 | head -30
for m in 1 -1; do DT_F4_MOSAIC=$m timeout 600 python tools/fused4_bench.py ${FRAMES:-1440} 2>&1 | grep -E "conv_[235] " | grep "fused4=2" | sed "s/^/F4_MOSAIC=$m /"; done | tee $O/bench.txt
