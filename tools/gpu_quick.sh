#!/bin/bash
# GPU box: selected tests + a short bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-quick}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout ${TMO:-1500} python -m pytest tests -m gpu -q -x ${KEXPR:+-k "$KEXPR"} --durations=5 > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
grep -E "^E  |^FAILED|Error" $O/pytest.txt | cut -c1-400 | head -30
if [ "${BENCH:-1}" = "1" ]; then timeout 900 python bench.py --no-cpu-baseline --no-extra --layer-report $O/bench_layers.txt 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_instrumented'], {k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"; fi
