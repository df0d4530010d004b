#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/out.txt
for i in 1 2; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"; done | tee -a $O/out.txt
