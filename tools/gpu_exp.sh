#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_$label.txt 2>$O/bench_$label.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s %8.1f frames/s %7.2f ms | ' % ('$label', d['value'], d['ms_per_step']) + ' '.join('%s %.2f' % (n.replace('conv_','').replace('wino_','w_'), k[n]['ms_per_step']) for n in ('conv1_direct','conv_fused','wino_input','wino_output','conv_gemm_s3','conv_igemm') if n in k))"
}
run base X=1
run nt0 MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_nt0.so
run base2 X=1
run nt0b MI355_DT_LIB=$R/tools/_probe_builds/libmi355_dt_nt0.so
