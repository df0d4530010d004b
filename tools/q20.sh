cd $GRAFT_REPO_ROOT; O=gpurun_out/q20; mkdir -p $O
for m in "1 1024" "0 1024" "1 256" "1 512"; do set -- $m; echo "== DT_S3_1X1=$1 MINK=$2"; DT_S3_1X1=$1 DT_S3_1X1_MINK=$2 timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c60-170; done
timeout 300 python tools/batch8_trace.py 8 200 0 2>&1 | grep WALL
