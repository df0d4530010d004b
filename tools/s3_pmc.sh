#!/bin/bash
# PMC passes on the split-bf16 GEMM micro-benchmark (each --pmc set in its own run, --kernel-trace only).  usage: s3_pmc.sh tag [binary suffix]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s3pmc}; SUF=${2:-}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o pmc -- $R/tools/micro/gemm_s3_bench$SUF 2 > $O/pmc_$n.log 2>&1
  for c in "$@"; do python $R/tools/rocprof_summary.py pmc $O/pmc_$n $c | grep -i "s3" | head -2 | cut -c1-30,88- | sed "s/^/$c /"; done
}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- $R/tools/micro/gemm_s3_bench$SUF 2 > $O/st.log 2>&1; python $R/tools/rocprof_summary.py stats $O/st | grep -i s3 | cut -c1-30,88-150
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run f TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum
find $O -name "*.csv" -size +20M -delete
