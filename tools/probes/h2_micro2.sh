cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/h2micro
( echo "== NT=2 8 waves timing"; S3_NT=2 timeout 300 tools/micro/gemm_s3_bench_t2 3
  echo "== NT=3 8 waves timing"; S3_NT=3 timeout 300 tools/micro/gemm_s3_bench_t2 3
  echo "== NT=2 4 waves"; S3_NT=2 S3_WAVES=4 timeout 300 tools/micro/gemm_s3_bench_w4 3
  echo "== NT=2 4 waves timing"; S3_NT=2 S3_WAVES=4 timeout 300 tools/micro/gemm_s3_bench_w4t 3
  echo "== NT=2 zero fill"; S3_NT=2 S3_FILL=zero timeout 300 tools/micro/gemm_s3_bench 3
) > gpurun_out/h2micro/out2.txt 2>&1
cat gpurun_out/h2micro/out2.txt
