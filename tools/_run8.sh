mkdir -p gpurun_out/r05h
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05h
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-oracle-check --no-roofline > $OUT/kt.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/kt.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/kt/kt_results.db 100 $MS 80 > $OUT/gd_kernel_stats.txt 2>&1
rm -rf $OUT/kt
rocprofv3 --kernel-trace --stats -d $OUT/ki -o ki -- python3 tools/itr_bench.py 384 64 10 > $OUT/ki.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/ki.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/ki/ki_results.db 250 $MS 80 > $OUT/itr_kernel_stats.txt 2>&1
rm -rf $OUT/ki
rocprofv3 --kernel-trace --stats -d $OUT/kv -o kv -- python3 tools/vqa_bench.py 480 32 10 > $OUT/kv.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/kv.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/kv/kv_results.db 220 $MS 80 > $OUT/vqa_kernel_stats.txt 2>&1
rm -rf $OUT/kv
head -3 $OUT/gd_kernel_stats.txt $OUT/itr_kernel_stats.txt $OUT/vqa_kernel_stats.txt
