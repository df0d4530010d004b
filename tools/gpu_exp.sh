#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python bench.py --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(k, {a:b for a,b in v.items() if a!='workload'}) for k,v in d['extra'].items()]" | tee $O/out.txt; tail -3 $O/err.txt
