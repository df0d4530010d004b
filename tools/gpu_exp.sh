#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "assoc or track" 2>&1 | tail -3 | tee -a $O/out.txt
for v in "" "" ; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --layer-report $O/layers_${v:-main}.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-main}', d['value'], d['ms_per_step'])"
  grep -E "^associate  |^decode_nms  |^conv_gemm_s3  |^conv_fused  |^conv1_direct  " $O/layers_${v:-main}.txt | cut -c1-75
done 2>&1 | tee -a $O/out.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -x -k "bench_size_48 or configs2_track_416_reference_default" 2>&1 | tail -3 | tee -a $O/out.txt
