#!/bin/bash
# A/B of the tile-order group width: time (bench) and FETCH_SIZE (one PMC pass) per setting
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O
for g in "$@"; do
  cd $R
  DT_TILE_GN=$g python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GN', $g, 'fps', round(d['value'],1), 'TF', round(d['roofline']['achieved'],2))"
  cd /tmp && export TMPDIR=/tmp
  DT_TILE_GN=$g timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gn$g -o pmc -- python $R/tools/pmc_probe.py 48 > $O/gn$g.log 2>&1
  python $R/tools/rocprof_summary.py pmc $O/gn$g FETCH_SIZE | grep conv_igemm | cut -c1-60,93- | head -4
  rm -rf $O/gn$g
done
