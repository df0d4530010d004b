cd $GRAFT_REPO_ROOT; O=gpurun_out/q14; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "wino or detector or conv2d or batch_invariance or tracker" > $O/t.txt 2>&1; tail -4 $O/t.txt
timeout 300 python tools/batch8_trace.py 8 200 0 2>&1 | grep WALL
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b8 -- python $R/tools/batch8_trace.py 8 100 0 > $R/$O/b8.log 2>&1
python $R/tools/rocprof_summary.py stats $R/$O/prof > $R/$O/batch8_kernel_stats.txt 2>&1; head -16 $R/$O/batch8_kernel_stats.txt
find $R/$O -name "*.csv" -size +5M -delete
