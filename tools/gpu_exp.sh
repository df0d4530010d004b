#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2 3; do timeout 300 python tools/c1_time.py 2>&1 | tail -1; done | tee $O/c1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "conv1 or detector or first_layer or extract" 2>&1 | tail -3 | tee -a $O/c1.txt
for i in 1 2; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"; done | tee -a $O/c1.txt
