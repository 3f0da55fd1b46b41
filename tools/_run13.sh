mkdir -p gpurun_out/r05m
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r05m/pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05m/pytest_gpu.log
tail -n 12 gpurun_out/r05m/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05m/smoke.log 2>&1; tail -n 2 gpurun_out/r05m/smoke.log
