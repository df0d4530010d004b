#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/h2_small; mkdir -p $O; cd $R
for B in ${BATCHES:-4 8 12 16 24}; do for E in "DT_H2_MINFRAMES=100" "DT_H2_MINFRAMES=4" "DT_H2_MINFRAMES=4 DT_S3_MINROWS=64" "DT_H2_MINFRAMES=4 DT_S3_MINROWS=256"; do
  env $E timeout 600 python bench.py --workload detect --batch $B --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('batch %3d %-42s' % ($B, '$E'), round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', {n:round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>0.03})"
done; done | tee $O/out.txt
