mkdir -p gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BASE=$GRAFT_REPO_ROOT/tools/_build/libevlm_base.so
O=gpurun_out/r05g/ab.txt
: > $O
for rep in 1 2; do
for lib in new base; do
  if [ $lib = base ]; then export EVLM_LIB=$BASE; else unset EVLM_LIB; fi
  echo "== rep $rep lib $lib" >> $O
  timeout 600 python3 tools/attn_bench.py 2>/dev/null | grep "fwd+bwd\|fused KD" >> $O
  timeout 600 python3 tools/attn_long_bench.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d['stream'])" >> $O
  timeout 600 python3 tools/itr_bench.py 384 64 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ITR', d['ms_per_step'])" >> $O
  timeout 600 python3 tools/vqa_bench.py 480 32 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('VQA', d['ms_per_step'])" >> $O
  python bench.py --no-cpu-baseline --no-oracle-check --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; print('GD', json.loads(sys.stdin.read())['ms_per_step'])" >> $O
done; done
cat $O
