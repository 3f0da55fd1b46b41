mkdir -p gpurun_out/r05l
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 2400 python -m pytest tests/test_step_gpu.py -x -q -m gpu -s -k "itr_384 or vqa_480 or captured_pruning" ) > gpurun_out/r05l/pytest_step.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05l/pytest_step.log
grep -n "gradient parity\|passed\|failed\|real\|Error" gpurun_out/r05l/pytest_step.log | cut -c1-900
