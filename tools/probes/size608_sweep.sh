#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/size608; mkdir -p $O; cd $R
for C in 1 2 4 8 16 24; do
  timeout 600 python bench.py --size 608 --clips $C --boxes 128 --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('608: clips %3d' % $C, round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', round(d['ms_per_step']/($C*30)*1e3,1), 'us/frame', {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>0.3})"
done | tee $O/out.txt
