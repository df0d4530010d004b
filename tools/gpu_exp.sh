#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv1_split or benched_shapes or detector_forward_vs_oracle_small" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "608_128" 2>&1 | tail -5
for m in 1 0; do
  DT_S3_CONV1=$m timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_c1_$m.txt 2>$O/bench_c1_$m.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DT_S3_CONV1=$m', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms; conv1', round(d['kernels']['conv1_direct']['ms_per_step'],3))"
done
for v in e0 e1; do echo "== gemm_s3_bench_$v"; timeout 300 tools/micro/gemm_s3_bench_$v 2>&1 | cut -c1-150; done
