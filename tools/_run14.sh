mkdir -p gpurun_out/r05n
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "teacher_recipe or long_sequence or streaming" > gpurun_out/r05n/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05n/pytest_ops.log
tail -n 25 gpurun_out/r05n/pytest_ops.log
for tq in 1 2; do
EVLM_ATTN_STREAM_TQ=$tq timeout 600 python3 tools/attn_long_bench.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('TQ=$tq', d['shape'], d['stream'])" > gpurun_out/r05n/attn_long_tq$tq.txt
cat gpurun_out/r05n/attn_long_tq$tq.txt
done
