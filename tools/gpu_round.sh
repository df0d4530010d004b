#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test suite, default bench line + per-layer
# table, rocprofv3 kernel stats of the same bench command, PMC passes (FETCH_SIZE,
# WRITE_SIZE separately; --kernel-trace only).  Outputs land in gpurun_out/<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu_full.txt 2>&1
  tail -25 $O/pytest_gpu_full.txt | tee $O/pytest_gpu.txt
  grep -E "^E  |^FAILED|Error" $O/pytest_gpu_full.txt | cut -c1-400 | head -60
fi
timeout 600 python bench.py --layer-report $O/bench_layers.txt 2>$O/bench.err | tail -1 | tee $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/prof_stats.log 2>&1
tail -1 $O/prof_stats.log
python $R/tools/rocprof_summary.py stats $O/prof_stats > $O/rocprof_kernel_stats.txt 2>&1
head -12 $O/rocprof_kernel_stats.txt
if [ "${SKIP_PMC:-0}" != "1" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$C -o pmc -- python $R/tools/pmc_probe.py 48 > $O/pmc_$C.log 2>&1
    tail -1 $O/pmc_$C.log
    python $R/tools/rocprof_summary.py pmc $O/pmc_$C $C > $O/rocprof_pmc_$C.txt 2>&1
    head -8 $O/rocprof_pmc_$C.txt
  done
  # MFMA-busy share and effective shader clock of every kernel INSIDE the step (one pass, both counters)
  timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_MFMA -o pmc -- python $R/tools/pmc_probe.py 48 > $O/pmc_MFMA.log 2>&1
  tail -1 $O/pmc_MFMA.log
  python $R/tools/rocprof_summary.py mfma $O/pmc_MFMA > $O/rocprof_pmc_MFMA.txt 2>&1
  head -14 $O/rocprof_pmc_MFMA.txt
fi
# the split-bf16 GEMM on its own: rate, error against float64, effective clock with / without the LDS fragment reads and the DMA
if [ -x $R/tools/micro/gemm_s3_bench ]; then
  (cd $R && for NT in 2 3; do S3_NT=$NT timeout 300 tools/micro/gemm_s3_bench; done > $O/gemm_s3_micro.txt 2>&1; cat $O/gemm_s3_micro.txt | cut -c1-170)
  (export S3_NT=2; bash $R/tools/s3_clock.sh - _a1 _a2 _a3 -zero -const > $O/gemm_s3_clock.txt 2>&1; export S3_NT=3; bash $R/tools/s3_clock.sh - -zero >> $O/gemm_s3_clock.txt 2>&1; cat $O/gemm_s3_clock.txt)
fi
# practical HBM ceilings of this box per traffic mix (read / write / copy / the transforms' mixes)
[ -x $R/tools/micro/hbm_rw ] && (cd $R && timeout 120 tools/micro/hbm_rw > $O/hbm_ceilings.txt 2>&1; cat $O/hbm_ceilings.txt)
# keep the merged payload small: drop raw per-dispatch CSVs above 20 MB
find $O -name "*.csv" -size +20M -delete
python $R/tools/make_traffic_json.py $O ${CLIPS:-48} > $O/traffic.json 2>/dev/null; cat $O/traffic.json | head -20
du -sh $O
