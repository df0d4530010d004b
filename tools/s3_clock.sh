#!/bin/bash
# effective shader clock of the GEMM micro-benchmark variants: GRBM_GUI_ACTIVE (summed over 8 XCDs) / kernel duration
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/s3clk; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
# a suffix "-zero" / "-const" runs the full kernel on all-zero / constant operands (S3_FILL): the clock the same instruction stream gets without data toggling
for SUF in "$@"; do
  [ "$SUF" = "-" ] && SUF=""
  FILL=""; BIN=$SUF
  case "$SUF" in -zero) FILL=zero; BIN="";; -const) FILL=const; BIN="";; esac
  S3_FILL=$FILL timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p$SUF -o pmc -- $R/tools/micro/gemm_s3_bench$BIN 2 > $O/p$SUF.log 2>&1
  python - $O/p$SUF "$SUF" <<'PY'
import csv, glob, sys, os
d, suf = sys.argv[1], sys.argv[2]
dur = {}; gui = {}; mf = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino_gemm_s3" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": gui[r["Dispatch_Id"]] = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES": mf[r["Dispatch_Id"]] = float(r["Counter_Value"])
ks = [k for k in gui if k in dur][2:]
if ks:
    us = sum(dur[k] for k in ks) / len(ks); cyc = sum(gui[k] for k in ks) / len(ks) / 8; m = sum(mf.get(k, 0) for k in ks) / len(ks) / 1024
    print("NT=%s variant '%s': %.0f us  %.2fM cycles  clock %.3f GHz  MFMA busy %.2fM cycles/SIMD (%.0f%%)" % (os.environ.get("S3_NT", "3"), suf, us, cyc / 1e6, cyc / us / 1e3, m / 1e6, 100 * m / cyc))
PY
done
