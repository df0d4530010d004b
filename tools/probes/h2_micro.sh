cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/h2micro
( echo "== NT=3"; S3_NT=3 timeout 600 tools/micro/gemm_s3_bench
  echo "== NT=2 NS=4"; S3_NT=2 timeout 600 tools/micro/gemm_s3_bench
  echo "== NT=2 NS=3"; S3_NT=2 timeout 600 tools/micro/gemm_s3_bench_v3 3
  echo "== NT=2 smallrows 20"; S3_NT=2 S3_SMALLROWS=20 timeout 300 tools/micro/gemm_s3_bench 3
  echo "== NT=2 smallrows 12"; S3_NT=2 S3_SMALLROWS=12 timeout 300 tools/micro/gemm_s3_bench 3
) > gpurun_out/h2micro/out.txt 2>&1
cat gpurun_out/h2micro/out.txt
