cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q8
for V in _a1 _a2 _a3 _a4 _a8 _a15; do echo "== ablation $V" | tee -a gpurun_out/q8/s4_ablate.txt
MI355_DT_LIB=$GRAFT_REPO_ROOT/object_tracking_amd/ablate/libmi355_dt_s4tt$V.so timeout 300 python tools/s4_timing.py conv_3 1440 2>&1 | grep -v "Native\|amdgpu.ids" | head -3 | tee -a gpurun_out/q8/s4_ablate.txt
done
