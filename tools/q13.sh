cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q13; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b8 -- python $R/tools/batch8_trace.py 8 100 0 > $O/b8.log 2>&1
grep WALL $O/b8.log
python $R/tools/rocprof_summary.py stats $O/prof > $O/batch8_kernel_stats.txt 2>&1; head -20 $O/batch8_kernel_stats.txt
find $O -name "*.csv" -size +5M -delete
