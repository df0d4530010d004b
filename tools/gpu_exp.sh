#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for a in "--steps 5 --warmup 2" "--steps 20 --warmup 2" "--steps 5 --warmup 12" "--steps 20 --warmup 10" "--steps 5 --warmup 2"; do
  timeout 600 python bench.py $a --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
done | tee $O/out.txt
