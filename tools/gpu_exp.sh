#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 600 python tools/diag_iou.py 2>&1 | grep -v "^Native" | tail -6
for v in d0 d1; do echo "== gemm_s3_bench_$v"; timeout 300 tools/micro/gemm_s3_bench_$v 2>&1 | cut -c1-150; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "split_bf16 or winograd or detector_forward or tracker or convlstm" 2>&1 | tail -8
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_$label.txt 2>$O/bench_$label.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s %8.1f frames/s %7.2f ms | ' % ('$label', d['value'], d['ms_per_step']) + ' '.join('%s %.2f' % (n.replace('conv_','').replace('wino_','w_'), k[n]['ms_per_step']) for n in ('conv1_direct','conv_fused','wino_input','wino_output','conv_gemm_s3','conv_igemm') if n in k))"
}
run base X=1
run base2 X=1
