#!/bin/bash
# GPU box: the split GEMM's row thresholds (DT_S3_MINROWS for the F(6x6) layers, DT_S3_REC_MINROWS for the recurrent step, DT_S3_1X1_MINROWS) against the clips per step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rec_minrows; mkdir -p $O; cd $R
for C in ${CLIPS:-2 4 8 12 24 36}; do for E in "DT_X=0" "DT_S3_REC_MINROWS=32" "DT_S3_REC_MINROWS=32 DT_S3_MINROWS=256" "DT_S3_REC_MINROWS=32 DT_S3_MINROWS=256 DT_S3_1X1_MINROWS=2048"; do
  env $E timeout 600 python bench.py --clips $C --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d %-70s' % ($C, '$E'), round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.3})"
done; done | tee $O/out.txt
