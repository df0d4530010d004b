cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_f4x4 or conv2_shape" > gpurun_out/q1/t_fused.txt 2>&1; tail -15 gpurun_out/q1/t_fused.txt
timeout 600 python tools/fused4_bench.py 1440 > gpurun_out/q1/fused4_bench.txt 2>&1; cat gpurun_out/q1/fused4_bench.txt | grep -v Native
timeout 900 python -m pytest tests/test_gpu_multi.py -q -k "four_ranks" > gpurun_out/q1/t_multi.txt 2>&1; grep -E "RANK|passed|failed" gpurun_out/q1/t_multi.txt | cut -c1-600
timeout 1200 python -m pytest tests/test_gpu_configs.py -q -k "reference_default" > gpurun_out/q1/t_defaults.txt 2>&1; grep -E "^E  |passed|failed" gpurun_out/q1/t_defaults.txt | cut -c1-700 | head -40
