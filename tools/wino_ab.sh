#!/bin/bash
# A/B of the Winograd GEMM launch parameters on the default bench workload (per-layer report).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wab; mkdir -p $O; cd $R
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --layer-report $O/layers_$n.txt 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value'],1), 'fps', round(d['ms_per_step'],2), 'ms')"
  grep -E "conv_igemm:(conv_9|conv_14|conv_19|conv_22|convlstm_step|convlstm_xproj) " $O/layers_$n.txt | awk '{printf "   %-30s %8.3f ms %7.2f TF\n",$1,$3,$4}'
}
run base
run cfg128 DT_WINO_CFG=0
run gn1 DT_WINO_GN=1
run gn4 DT_WINO_GN=4
run gn0 DT_WINO_GN=0
