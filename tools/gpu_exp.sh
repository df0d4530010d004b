#!/bin/bash
# scratch experiment runner on the GPU box (via gpurun): edit freely between calls; outputs land in gpurun_out/<tag>/
#   A/B of one translation unit: tools/build_variant.sh <name> <unit.hip> <flags>  ->  MI355_DT_LIB=tools/_probe_builds/libmi355_dt_<name>.so
#   bit-comparison of two builds: tools/bitcmp.py save a.npz (per build), then tools/bitcmp.py cmp a.npz b.npz
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-exp}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R

for i in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --layer-report $O/layers.txt 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
done | tee $O/out.txt
