#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for g in "" "--events-all" "" "--events-all"; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra $g --layer-report $O/layers${g}.txt 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${g:-default}', d['value'], d['ms_per_step'], r['frac'], r['events'], {k:round(v['ms_per_step'],2) for k,v in r['families'].items()}, round(r['transforms']['ms_per_step'],2))"
  tail -3 $O/err.txt
done 2>&1 | tee $O/out.txt
