mkdir -p gpurun_out/r05o
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "teacher_recipe" > gpurun_out/r05o/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05o/pytest_ops.log
tail -n 4 gpurun_out/r05o/pytest_ops.log
O=gpurun_out/r05o/ab.txt; : > $O
for rep in 1 2; do
for v in "-" "EVLM_NO_KD_RECIPE=1" "EVLM_ATTN_STREAM_TQ=1"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  i=$(env $e timeout 600 python3 tools/itr_bench.py 384 64 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['launch'], d['losses[total,itc,itm,kd,lagrangian]'])")
  q=$(env $e timeout 600 python3 tools/vqa_bench.py 480 32 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['launch'], d['losses[total,answer,kd,lagrangian]'])")
  echo "rep $rep [$v] ITR $i | VQA $q" >> $O
done; done
cat $O
timeout 2400 python -m pytest tests/test_step_gpu.py -x -q -m gpu -s -k "itr_384 or vqa_480 or captured_pruning or pruning_step or itr_trainer" > gpurun_out/r05o/pytest_step.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05o/pytest_step.log
grep -n "gradient parity\|passed\|failed\|Error" gpurun_out/r05o/pytest_step.log | cut -c1-400
