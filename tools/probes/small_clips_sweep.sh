#!/bin/bash
# GPU box: the tracking step at 1-6 clips against the tile-count threshold of the Winograd form (DT_WINO_MINT; default 32 F(6x6) / 64 F(4x4) tiles)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/small_clips; mkdir -p $O; cd $R
for C in ${CLIPS:-1 2 3 4 6}; do for E in "DT_X=0" "DT_WINO_MINT=32" "DT_WINO_MINT=16" "DT_WINO_MINT=8"; do
  env $E timeout 600 python bench.py --clips $C --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('clips %3d %-20s' % ($C, '$E'), round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.2})"
done; done | tee $O/out.txt
