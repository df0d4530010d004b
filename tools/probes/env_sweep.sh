#!/bin/bash
# GPU box: the bench line under a list of environments on ONE box:  tools/probes/env_sweep.sh <tag> "VAR=a" "VAR=b VAR2=c" ...   (first = baseline, repeated at the end)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-sweep}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for E in "$@" "$1"; do
  env $E timeout 600 python bench.py --steps ${STEPS:-4} --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-44s' % '$E', round(d['value'],1), round(d['ms_per_step'],2), {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>1})"
done | tee $O/out.txt
