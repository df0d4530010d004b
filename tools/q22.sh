cd $GRAFT_REPO_ROOT; O=gpurun_out/q22; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "split_bf16 or detector" > $O/t.txt 2>&1; tail -3 $O/t.txt
for m in 512 256; do echo "== MINK=$m"; DT_S3_1X1_MINK=$m timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 --layer-report $O/layers$m.txt 2>/dev/null | tail -1 | cut -c60-170; grep "wino_output \|wino_output:conv_6\|wino_output:conv_9\|wino_output:conv_14\|conv_gemm_s3:conv_7\|conv_igemm:conv_7" $O/layers$m.txt; done
