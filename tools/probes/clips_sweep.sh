#!/bin/bash
# GPU box: per-frame time of the tracking step against the number of clips per step (does a smaller working set run faster per frame?)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/clips_sweep; mkdir -p $O; cd $R
for C in 48 6 12 24 36 48 72 96; do
  timeout 600 python bench.py --clips $C --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d' % $C, round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', round(d['ms_per_step']/($C*30)*1e3,2), 'us/frame', {n:round(v['ms_per_step']/($C*30)*1e3,2) for n,v in k.items() if v['ms_per_step']>0.5})"
done | tee $O/out.txt
