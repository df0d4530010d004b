cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q11
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_f4x4 or conv2_shape" > gpurun_out/q11/t_fused.txt 2>&1; tail -3 gpurun_out/q11/t_fused.txt
for L in conv_3 conv_2; do
MI355_DT_LIB=$GRAFT_REPO_ROOT/object_tracking_amd/ablate/libmi355_dt_s4tt.so timeout 300 python tools/s4_timing.py $L 1440 2>&1 | grep -v "Native\|amdgpu.ids" | head -4 | tee -a gpurun_out/q11/s4_timing.txt
done
timeout 600 python tools/fused4_bench.py 1440 2>&1 | grep "w4s=2" | tee gpurun_out/q11/fused4_bench.txt
