cd $GRAFT_REPO_ROOT; O=gpurun_out/q17; mkdir -p $O
timeout 3000 python -m pytest tests -q -x -m gpu > $O/t.txt 2>&1; tail -8 $O/t.txt
timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 --layer-report $O/layers.txt 2>$O/bench.err | tail -1 | cut -c1-300
grep "conv_gemm_s3\|conv_igemm \|wino_input \|wino_output \|conv_fused " $O/layers.txt
