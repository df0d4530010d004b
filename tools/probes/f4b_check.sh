#!/bin/bash
# GPU box: parity tests of the fused F(4x4) kernels (bf16-split kernel by default, DT_F4B=0: the fp32 kernel) + per-layer timing at the bench batch
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-f4b}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
DT_F4B=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or conv2_shape or non_square" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
grep -E "^E  |^FAILED|Error" $O/pytest.txt | cut -c1-300 | head -30
for f in 1 0; do DT_F4B=$f timeout 600 python tools/fused4_bench.py ${FRAMES:-1440} 2>&1 | grep -E "conv_[235] " | grep "fused4=2" | sed "s/^/F4B=$f /"; done | tee $O/bench.txt
