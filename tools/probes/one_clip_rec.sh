#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/one_clip; mkdir -p $O; cd $R
for C in 1 2; do for E in "DT_WINO_MINT=16" "DT_WINO_MINT=16 DT_S3_REC_MINROWS=16" "DT_WINO_MINT=16 DT_S3_REC_MINROWS=16 DT_S3_HALF=1" "DT_WINO_MINT=16 DT_H2_MINFRAMES=40"; do
  env $E timeout 600 python bench.py --clips $C --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d %-60s' % ($C, '$E'), round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.2})"
done; done | tee $O/out.txt
