cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q10
for V in _a32 _a64; do echo "== $V" | tee -a gpurun_out/q10/s4_probe2.txt
MI355_DT_LIB=$GRAFT_REPO_ROOT/object_tracking_amd/ablate/libmi355_dt_s4tt$V.so timeout 300 python tools/s4_timing.py conv_3 1440 2>&1 | grep -v "Native\|amdgpu.ids" | head -4 | tee -a gpurun_out/q10/s4_probe2.txt
done
