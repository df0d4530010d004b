cd $GRAFT_REPO_ROOT; O=gpurun_out/q23; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "convlstm or tracker or track or configs2 or graph_replay" > $O/t.txt 2>&1; tail -5 $O/t.txt
for m in 512 0; do echo "== REC_MINROWS=$m"; DT_S3_REC_MINROWS=$m timeout 900 python bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 --layer-report $O/layers$m.txt 2>/dev/null | tail -1 | cut -c60-170; grep "convlstm_step" $O/layers$m.txt; done
