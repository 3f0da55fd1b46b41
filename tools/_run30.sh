mkdir -p gpurun_out/r05ac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r05ac/pytest_all.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05ac/pytest_all.log
tail -n 4 gpurun_out/r05ac/pytest_all.log | cut -c1-300
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 | cut -c1-300
