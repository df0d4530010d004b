#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "conv1_split or split_bf16 or winograd or fused or convlstm or detector_forward or tracker" 2>&1 | tail -8
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_$label.txt 2>$O/bench_$label.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s %8.1f frames/s %7.2f ms | ' % ('$label', d['value'], d['ms_per_step']) + ' '.join('%s %.2f' % (n.replace('conv_','').replace('wino_','w_'), k[n]['ms_per_step']) for n in ('conv1_direct','conv_fused','wino_input','wino_output','conv_gemm_s3','conv_igemm') if n in k))"
}
run base X=1
run ws0 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_ws0.so
run e0 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_e0.so
run mink128 DT_S3_MINK=128
run c1pf2 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_c1pf2.so
run c1f32 DT_S3_CONV1=0
run us2 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_us2.so
run us3 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_us3.so
run us4 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_us4.so
run base2 X=1
