mkdir -p gpurun_out/r05f
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r05f/pytest_ops.log 2>&1
echo "pytest rc $?" >> gpurun_out/r05f/pytest_ops.log
tail -n 4 gpurun_out/r05f/pytest_ops.log
timeout 600 python3 tools/attn_long_bench.py > gpurun_out/r05f/attn_long_bench.jsonl 2> gpurun_out/r05f/attn_long.err
timeout 600 python3 tools/attn_bench.py > gpurun_out/r05f/attn_bench.txt 2> gpurun_out/r05f/attn_bench.err
timeout 600 python3 tools/itr_bench.py 384 64 10 > gpurun_out/r05f/itr.json 2> gpurun_out/r05f/itr.err
timeout 600 python3 tools/vqa_bench.py 480 32 10 > gpurun_out/r05f/vqa.json 2> gpurun_out/r05f/vqa.err
bash tools/ab_step.sh 2 - > gpurun_out/r05f/gd_step.txt 2>&1
cat gpurun_out/r05f/attn_long_bench.jsonl gpurun_out/r05f/itr.json gpurun_out/r05f/vqa.json gpurun_out/r05f/gd_step.txt
grep "ViT 64x12x197 fwd+bwd\|cross 256" gpurun_out/r05f/attn_bench.txt
