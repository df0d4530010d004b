#!/bin/bash
# GPU box: DT_H2_MINFRAMES (fp16 form + publications from this many frames per forward) against the detector batch; then the tracking step at a few clip counts (new default thresholds)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/h2_minframes; mkdir -p $O; cd $R
for B in 8 12 16 24 32; do for M in 32 8; do
  DT_H2_MINFRAMES=$M timeout 600 python bench.py --workload detect --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('batch %3d h2_minframes %3d' % ($B, $M), round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', {n:round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>0.05})"
done; done | tee $O/out.txt
for C in 2 4 8 12 24 48; do
  timeout 600 python bench.py --clips $C --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d' % $C, round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.3})"
done | tee -a $O/out.txt
