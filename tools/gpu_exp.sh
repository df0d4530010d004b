#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in 256 128 256 128; do
  DT_S3_1X1_MINK=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --layer-report $O/layers_$v.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mink$v', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
  grep -E "conv_4 |conv_23 " $O/layers_$v.txt | cut -c1-62
done 2>&1 | tee $O/out.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | tee -a $O/out.txt
