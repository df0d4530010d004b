cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q2
for L in conv_3 conv_6 conv_2; do
MI355_DT_LIB=$GRAFT_REPO_ROOT/object_tracking_amd/ablate/libmi355_dt_s4tt.so timeout 300 python tools/s4_timing.py $L 1440 2>&1 | grep -v Native | tee -a gpurun_out/q2/s4_timing.txt
done
timeout 900 python -m pytest tests/test_gpu_multi.py -q -k "four_ranks" > gpurun_out/q2/t_multi.txt 2>&1; grep -E "RANK|passed|failed" gpurun_out/q2/t_multi.txt | cut -c1-600
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -k "reference_default" > gpurun_out/q2/t_defaults.txt 2>&1; grep -E "^E  |passed|failed" gpurun_out/q2/t_defaults.txt | cut -c1-900 | head -40
ls gpurun_out/*.json
