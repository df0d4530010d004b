#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for v in pf0 "" pf0 ""; do
  L=""; [ -n "$v" ] && L=$R/tools/_probe_builds/libmi355_dt_$v.so
  MI355_DT_LIB=$L timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra --layer-report $O/layers_${v:-main}.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-main}', d['value'], d['ms_per_step'])"
  grep -E "^wino_input  |^wino_input:conv_(6|9|14|19|22) |wino_input:convlstm" $O/layers_${v:-main}.txt | cut -c1-80
done 2>&1 | tee $O/out.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "wino or split or conv2d or tracker" 2>&1 | tail -3 | tee -a $O/out.txt
