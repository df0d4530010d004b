#!/bin/bash
# GPU box: conv_2 / conv_3 / conv_5 per-layer times of the direct fp16-form kernel under library variants (MI355_DT_LIB), inside the bench step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c3ab2; mkdir -p $O; cd $R
for rep in 1 2; do for V in "" "$@"; do
  LIB=""; [ -n "$V" ] && LIB=$R/tools/_probe_builds/libmi355_dt_$V.so
  MI355_DT_LIB=$LIB timeout 600 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra --layer-report $O/l_$V.txt 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$V]', round(d['value']), round(d['ms_per_step'],2), end=' ')"
  grep -E "conv_direct_h2:" $O/l_$V.txt | awk '{printf "%s %.2f  ", $1, $3} END {print ""}'
done; done | tee $O/out.txt
