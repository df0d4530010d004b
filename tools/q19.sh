cd $GRAFT_REPO_ROOT; O=gpurun_out/q19; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "split_bf16 or detector" > $O/t.txt 2>&1; tail -12 $O/t.txt
timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 --layer-report $O/layers.txt 2>$O/bench.err | tail -1 | cut -c1-200
grep "conv_gemm_s3\|conv_igemm\|wino_input \|wino_output\|conv_fused " $O/layers.txt | grep -v "convlstm_step"
