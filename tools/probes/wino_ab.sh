#!/bin/bash
# A/B runs on the default bench workload (per-layer report): usage  wino_ab.sh name ENV=.. -- name ENV=.. ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wab; mkdir -p $O; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_$n.txt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms')"
  grep -E "conv_igemm:(conv_2|conv_3|conv_6|conv_7|conv_9|conv_10|conv_14|conv_19|conv_22|convlstm_step|convlstm_xproj) " $O/layers_$n.txt | awk '{printf "   %-30s %8.3f ms %7.2f TF\n",$1,$3,$4}'
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run "${args[@]}"
