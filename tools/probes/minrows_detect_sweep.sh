#!/bin/bash
# GPU box: DT_S3_MINROWS against the detector batch (frames per forward), fp16 form (>= 32 frames)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/minrows_detect; mkdir -p $O; cd $R
for B in ${BATCHES:-32 48 64 96 128 192}; do for M in 2048 1024 512 256 128; do
  DT_S3_MINROWS=$M timeout 600 python bench.py --workload detect --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('batch %3d minrows %4d' % ($B, $M), round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', {n:round(v['ms_per_step'],3) for n,v in k.items() if v['ms_per_step']>0.05})"
done; done | tee $O/out.txt
