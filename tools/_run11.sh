mkdir -p gpurun_out/r05k
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "lagrangian or l0" > gpurun_out/r05k/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05k/pytest_ops.log
tail -n 15 gpurun_out/r05k/pytest_ops.log
timeout 2400 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "itr or vqa or pruning or pruned or l0 or stop_prune" > gpurun_out/r05k/pytest_step.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05k/pytest_step.log
tail -n 5 gpurun_out/r05k/pytest_step.log
O=gpurun_out/r05k/ab.txt; : > $O
for rep in 1 2; do
for v in "-" "EVLM_NO_FUSED_LAGRANGIAN=1" "EVLM_NO_FUSED_LAGRANGIAN=1 EVLM_NO_GATE_SLOTS=1"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  i=$(env $e timeout 600 python3 tools/itr_bench.py 384 64 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['launch'])")
  q=$(env $e timeout 600 python3 tools/vqa_bench.py 480 32 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['launch'])")
  echo "rep $rep [$v] ITR $i | VQA $q" >> $O
done; done
cat $O
