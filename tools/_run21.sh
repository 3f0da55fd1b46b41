mkdir -p gpurun_out/r05v
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or layer_norm or hidden_kd or ln_ or fused_hidden" > gpurun_out/r05v/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05v/pytest_ops.log
tail -n 5 gpurun_out/r05v/pytest_ops.log | cut -c1-300
O=gpurun_out/r05v/ln.txt; : > $O
for v in "-" "EVLM_LN_BWD_PHASED=1"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  echo "== [$v]" >> $O
  env $e timeout 300 python3 tools/ln_bench.py 2>/dev/null | grep ln_bwd >> $O
done
cat $O
O=gpurun_out/r05v/ab.txt; : > $O
for rep in 1 2 3; do
for v in "-" "EVLM_LN_BWD_PHASED=1"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  b=$(env $e timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "rep $rep [$v] GD $b" >> $O
done; done
cat $O
