#!/bin/bash
# copy the evidence of one tools/gpu_round.sh call (gpurun_out/<tag>/) into profiles/ under the round's names:
#   tools/collect_profiles.sh <tag> <rNN>
set -e
cd "$(dirname "$0")/.."
T=gpurun_out/$1; R=$2; P=profiles
python tools/make_traffic_json.py $T 48 > $P/${R}_traffic.json
cp $T/bench.json $P/${R}_bench.json
cp $T/bench_layers.txt $P/${R}_bench_layers.txt
cp $T/rocprof_kernel_stats.txt $P/${R}_rocprof_kernel_stats.txt
cp $T/pytest_gpu.txt $P/${R}_pytest_gpu.txt
{ echo "# FETCH_SIZE (KiB; gfx950: multiply by the in-run calibration ~2.0, ${R}_traffic.json) -- rocprofv3 --kernel-trace --pmc FETCH_SIZE over tools/pmc_probe.py 48 (one head-calibration forward + one measured step)"; cat $T/rocprof_pmc_FETCH_SIZE.txt
  echo; echo "# WRITE_SIZE (KiB) -- separate pass"; cat $T/rocprof_pmc_WRITE_SIZE.txt
  echo; echo "# effective shader clock and MFMA-busy share of every kernel INSIDE the step -- one pass --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES (tools/rocprof_summary.py mfma)"; cat $T/rocprof_pmc_MFMA.txt; } > $P/${R}_rocprof_pmc.txt
[ -f $T/gemm_s3_micro.txt ] && cp $T/gemm_s3_micro.txt $P/${R}_gemm_s3_micro.txt
[ -f $T/gemm_s3_clock.txt ] && cp $T/gemm_s3_clock.txt $P/${R}_gemm_s3_clock.txt
[ -f $T/hbm_ceilings.txt ] && { echo "# tools/micro/hbm_rw (16 B per lane, grid-stride, nontemporal): what a plain streaming kernel reaches on the box, per traffic mix"; cat $T/hbm_ceilings.txt; } > $P/${R}_hbm_ceilings.txt
for f in gpurun_out/parity_${R}_*.json gpurun_out/parity_${R}_*.txt; do [ -f $f ] && cp $f $P/; done
ls $P | grep "^${R}_\|parity_${R}"
