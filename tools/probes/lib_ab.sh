#!/bin/bash
# GPU box: the bench line under library variants (MI355_DT_LIB) and / or environments, alternating on ONE box:
#   tools/probes/lib_ab.sh <tag> "<variant or ->[:ENV=val]" ...      e.g.  lib_ab.sh ns  "-" "ns3" "-:DT_S3_HALF=-1"
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do for S in "$@"; do
  V=${S%%:*}; E=""; [ "$S" != "$V" ] && E=${S#*:}
  LIB=""; [ "$V" != "-" ] && LIB=$R/tools/_probe_builds/libmi355_dt_$V.so
  env MI355_DT_LIB=$LIB $E timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[$S]', round(d['value']), round(d['ms_per_step'],2), {n:round(v['ms_per_step'],2) for n,v in k.items() if v['ms_per_step']>1})"
done; done | tee $O/out.txt
