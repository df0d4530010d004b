#!/bin/bash
# PMC passes on one layer kernel (each --pmc set in its own run, --kernel-trace only).  usage: fused4_pmc.sh conv_3 480 tag
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
L=${1:-conv_3}; N=${2:-480}; TAG=${3:-f4}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o pmc -- python $R/tools/fused4_probe.py $L $N > $O/pmc_$n.log 2>&1
  for c in "$@"; do python $R/tools/rocprof_summary.py pmc $O/pmc_$n $c | grep -i "fused\|wino\|igemm\|conv3" | head -4 | cut -c1-40,88- | sed "s/^/$c /"; done
}
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run d SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run e SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS
run f TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum
find $O -name "*.csv" -size +20M -delete
