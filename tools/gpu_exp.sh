#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1; do timeout 900 python bench.py --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k, {a:b for a,b in v.items() if a not in ('workload','plain','graphs')}) for k,v in d['extra'].items()]"; done | tee $O/out.txt
timeout 600 python -m pytest tests/test_gpu_multi.py -q -x -k "partitions" 2>&1 | tail -2
