#!/bin/bash
# GPU box: wino4b_fused.hip -- parity tests, per-layer timing, phase timing of the -DDT_B4_TIMING variants named in $VARS
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-b4}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export DT_F4B=1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or conv2_shape or non_square" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
grep -E "^E  |^FAILED|Error" $O/pytest.txt | cut -c1-300 | head -30
fi
timeout 600 python tools/fused4_bench.py ${FRAMES:-1440} 2>&1 | grep -E "conv_[235] " | grep "fused4=2" | tee $O/bench.txt
for v in ${VARS:-tt0}; do for L in ${LAYERS:-conv_2}; do echo "== variant $v"; MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_b4$v.so timeout 300 python tools/b4_timing.py $L ${FRAMES:-1440} 2>&1 | grep -v "amdgpu.ids\|Warning\|wv, sel\|ret = ret\|sel\[" | head -${LINES:-5}; done; done | tee $O/timing.txt
