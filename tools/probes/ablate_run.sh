#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in "$@"; do
  L=$R/object_tracking_amd/libmi355_dt.so; [ "$m" != "0" ] && L=$R/tools/_probe_builds/libmi355_dt_abl$m.so
  MI355_DT_LIB=$L python bench.py --no-cpu-baseline --clips 24 --steps 3 --warmup 1 --layer-report /tmp/l_$m.txt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate', '$m', 'conv TF', round(d['roofline']['achieved'],2))"
  grep -E "conv_19 |conv_2 |conv_3 " /tmp/l_$m.txt | awk '{printf "   %s %s\n", $1, $4}'
done
