#!/bin/bash
# GPU box: the bench line under two environments on ONE box, alternating:  tools/ab_env.sh <tag> "VAR=a" "VAR=b" [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-ab}; A="$2"; B="$3"; N=${4:-2}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in $(seq 1 $N); do for E in "$A" "$B"; do
  env $E timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$E', round(d['value'],1), round(d['ms_per_step'],2), {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>1})"
done; done | tee $O/out.txt
