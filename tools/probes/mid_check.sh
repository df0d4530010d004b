#!/bin/bash
# GPU box: detector forward at a list of batches under a list of environments (ENVS: space-separated, ':' joins several variables of one environment), then the tracking step at CLIPS
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/mid_check; mkdir -p $O; cd $R
for B in ${BATCHES:-8 16 24 32 64}; do for E in ${ENVS:-DT_X=0}; do
  env ${E//:/ } timeout 600 python bench.py --workload detect --batch $B --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('batch %3d %-36s' % ($B, '$E'), round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', {n:round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>0.03})"
done; done | tee $O/out.txt
for C in ${CLIPS:-1 8 48}; do
  timeout 600 python bench.py --clips $C --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d' % $C, round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.2})"
done | tee -a $O/out.txt
