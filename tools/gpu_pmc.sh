#!/bin/bash
# PMC passes for the conv kernel (each --pmc set in its own run, --kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(gpu-agent|Name|[A-Z_0-9]+)\s*$|Name:" | grep -E "MFMA|GRBM_GUI|SQ_BUSY|SQ_WAVE_CYC|SQ_WAIT|SQ_ACTIVE_INST|LDS_BANK|LDS_IDX|SQ_INSTS_VALU_MFMA|MfmaUtil|TCC_HIT|TCC_MISS|TCC_EA0_RDREQ|TCC_REQ" | sort -u | head -60 > $O/counters_available.txt
run() { # name, counters...
  n=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o pmc -- python $R/tools/pmc_probe.py ${CLIPS:-48} > $O/pmc_$n.log 2>&1
  tail -1 $O/pmc_$n.log
  for c in "$@"; do python $R/tools/rocprof_summary.py pmc $O/pmc_$n $c | head -6 > $O/rocprof_pmc_$c.txt; done
}
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
find $O -name "*.csv" -size +20M -delete
cat $O/rocprof_pmc_*.txt | cut -c1-60,93-
