cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q6
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or conv2" > gpurun_out/q6/t_fused.txt 2>&1; tail -3 gpurun_out/q6/t_fused.txt
for L in conv_3 conv_2; do
MI355_DT_LIB=$GRAFT_REPO_ROOT/object_tracking_amd/ablate/libmi355_dt_s4tt.so timeout 300 python tools/s4_timing.py $L 1440 2>&1 | grep -v "Native\|amdgpu.ids" | head -4 | tee -a gpurun_out/q6/s4_timing.txt
done
timeout 600 python bench.py --no-cpu-baseline --no-extra --layer-report gpurun_out/q6/bench_layers.txt 2>gpurun_out/q6/bench.err | tail -1 > gpurun_out/q6/bench.json; python -c "
import json; d=json.load(open('gpurun_out/q6/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_whole_conv_path']); print({k:round(v['ms_per_step'],2) for k,v in d['kernels'].items()})"
grep "conv_fused" gpurun_out/q6/bench_layers.txt
