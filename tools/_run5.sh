mkdir -p gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or hidden_state or fork" > gpurun_out/r05e/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05e/pytest_ops.log
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "benchmarked or gd_step or itr_384 or pipelined_teacher" > gpurun_out/r05e/pytest_step.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05e/pytest_step.log
tail -n 4 gpurun_out/r05e/pytest_ops.log gpurun_out/r05e/pytest_step.log
bash tools/ab_step.sh 2 - EVLM_NO_FUSED_HIDDEN_KD=1 > gpurun_out/r05e/ab_hidden_kd.txt 2>&1
cat gpurun_out/r05e/ab_hidden_kd.txt
O=gpurun_out/r05e/cu_mask.txt
timeout 300 python3 tools/cu_mask_probe.py --mode map > $O 2> gpurun_out/r05e/cu_mask_map.err
for rep in 1 2; do
timeout 300 python3 tools/cu_mask_probe.py --mode joint >> $O 2>> gpurun_out/r05e/cu_mask.err
timeout 300 python3 tools/cu_mask_probe.py --mode joint --serial-text >> $O 2>> gpurun_out/r05e/cu_mask.err
timeout 300 python3 tools/cu_mask_probe.py --mode two >> $O 2>> gpurun_out/r05e/cu_mask.err
timeout 300 python3 tools/cu_mask_probe.py --mode two --serial-text >> $O 2>> gpurun_out/r05e/cu_mask.err
done
for tc in 32 64 96 128; do for lay in block striped; do
timeout 300 python3 tools/cu_mask_probe.py --mode masked --serial-text --teacher-cus $tc --layout $lay >> $O 2>> gpurun_out/r05e/cu_mask.err
done; done
cat $O
