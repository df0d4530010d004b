#!/bin/bash
# FETCH_SIZE + time for alternative library builds (MI355_DT_LIB)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/ab; mkdir -p $O
for n in "$@"; do
  L=$R/tools/_probe_builds/libmi355_dt_$n.so
  cd $R
  MI355_DT_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', 'fps', round(d['value'],1), 'TF', round(d['roofline']['achieved'],2))"
  cd /tmp && export TMPDIR=/tmp
  MI355_DT_LIB=$L timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$n -o pmc -- python $R/tools/pmc_probe.py 48 > $O/$n.log 2>&1
  python $R/tools/rocprof_summary.py pmc $O/$n FETCH_SIZE | grep conv_igemm | cut -c1-60,93- | head -3
  rm -rf $O/$n
done
