#!/bin/bash
# Round profile bundle (run on the GPU box from the repo root): default bench line, rocprofv3 kernel stats of the bench command,
# the PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) summarised per kernel, the secondary configurations, the launch-contract
# dry runs and the soak runs.  Only small summaries are kept.
#   EVLM_COMMIT=<sha> OUT=gpurun_out/r06p tools/profile_round.sh
set -u
OUT=${OUT:-gpurun_out/r06p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_1gpu.json
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-oracle-check --no-roofline > $OUT/kt.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/kt.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/kt/kt_results.db 100 $MS 80 > $OUT/kernel_stats.txt 2>&1
python3 tools/rocpd_stats.py $OUT/kt/kt_results.db 1 $OUT/kernel_stats.csv 5 > /dev/null 2>&1
grep -o "{\"metric.*" $OUT/kt.log > $OUT/bench_under_rocprof.json
rm -rf $OUT/kt
P="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-oracle-check --no-graph"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf -o pf -- $P > $OUT/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pw -o pw -- $P > $OUT/pw.log 2>&1
python3 tools/pmc_traffic.py $OUT/pf/pf_results.db $OUT/pw/pw_results.db $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
rm -rf $OUT/pf $OUT/pw
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pm -o pm -- $P > $OUT/pm.log 2>&1
python3 tools/pmc_mfma.py $OUT/pm/pm_results.db $OUT/pmc_mfma.json > $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/pm
python3 tools/xattn_bench.py 2>&1 | grep composite_us > $OUT/xattn_bench.jsonl
python3 tools/gemm_bench.py --lib 2>&1 | grep -v amdgpu.ids > $OUT/gemm_vs_library.txt
python3 tools/find_small_ops.py 2>&1 | grep -v amdgpu.ids > $OUT/aten_launching_calls.txt
python3 tools/attn_long_bench.py 2>/dev/null > $OUT/attn_long_bench.jsonl
# configs[2] / [3]: captured student steps on one GPU, their kernel tables, the N > 1 form on a one-rank RCCL group
python3 tools/itr_bench.py 384 64 10 2>/dev/null > $OUT/itr_step.jsonl
EVLM_FORCE_REDUCE=1 python3 tools/itr_bench.py 384 64 10 2>/dev/null | grep "^{" >> $OUT/itr_step.jsonl
python3 tools/vqa_bench.py 480 32 10 2>/dev/null > $OUT/vqa_step.jsonl
EVLM_FORCE_REDUCE=1 python3 tools/vqa_bench.py 480 32 10 2>/dev/null | grep "^{" >> $OUT/vqa_step.jsonl
# round 6: the same steps under the stock training-mode dropout (student BERT p = 0.1), and over an epoch's worth of batches in
# the reference's own padding (random real text lengths / answer counts) fed through bucket padding
python3 tools/itr_bench.py 384 64 10 --dropout 0.1 2>/dev/null | grep "^{" >> $OUT/itr_step.jsonl
python3 tools/vqa_bench.py 480 32 10 --dropout 0.1 2>/dev/null | grep "^{" >> $OUT/vqa_step.jsonl
python3 tools/itr_bench.py 384 64 --ragged 200 2>/dev/null | grep "^{" > $OUT/itr_ragged.jsonl
python3 tools/itr_bench.py 384 64 --ragged 200 --dropout 0.1 2>/dev/null | grep "^{" >> $OUT/itr_ragged.jsonl
python3 tools/vqa_bench.py 480 32 --ragged 200 2>/dev/null | grep "^{" > $OUT/vqa_ragged.jsonl
python3 tools/vqa_bench.py 480 32 --ragged 200 --dropout 0.1 2>/dev/null | grep "^{" >> $OUT/vqa_ragged.jsonl
python3 tools/ln_fwd_pair_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/ln_fwd_pair.txt
# round 6: the long-sequence attention backward - per-kernel durations of one ViT layer at 577 / 901 tokens (new forms, then the
# forms they replace), and the SQ counter passes of the 577-token layer without / with the teacher's recipe
( bash tools/attn_long_kernels.sh; PROBE_ARGS=--kd bash tools/attn_long_kernels.sh
  export EVLM_ATTN_DQ_NO_BATCH=1 EVLM_ATTN_DKV_NO_STREAM=1
  echo "-- round 4 / 5 forms (EVLM_ATTN_DQ_NO_BATCH=1 EVLM_ATTN_DKV_NO_STREAM=1)"
  bash tools/attn_long_kernels.sh; PROBE_ARGS=--kd bash tools/attn_long_kernels.sh ) 2>&1 | grep -v "^$" > $OUT/attn_long_kernels.txt
bash tools/attn_long_pmc.sh 64 577 > $OUT/attn_long_pmc_plain.txt 2>&1
bash tools/attn_long_pmc.sh 64 577 --kd > $OUT/attn_long_pmc_kd.txt 2>&1
( export EVLM_ATTN_DQ_NO_BATCH=1 EVLM_ATTN_DKV_NO_STREAM=1; bash tools/attn_long_pmc.sh 64 577 > $OUT/attn_long_pmc_plain_old_forms.txt 2>&1 )
rocprofv3 --kernel-trace --stats -d $OUT/ki -o ki -- python3 tools/itr_bench.py 384 64 10 > $OUT/ki.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/ki.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/ki/ki_results.db 250 $MS 70 > $OUT/itr384_kernel_stats.txt 2>&1
rm -rf $OUT/ki
rocprofv3 --kernel-trace --stats -d $OUT/kv -o kv -- python3 tools/vqa_bench.py 480 32 10 > $OUT/kv.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' $OUT/kv.log | head -1 | grep -o '[0-9.]*$')
python3 tools/replay_window_stats.py $OUT/kv/kv_results.db 220 $MS 70 > $OUT/vqa480_kernel_stats.txt 2>&1
rm -rf $OUT/kv
# configs[4], region recipe, rerank loop
python3 tools/pruned_inference_sweep.py 2>/dev/null > $OUT/pruned_inference.jsonl
python3 tools/pruned_inference_sweep.py masked 2>/dev/null > $OUT/masked_dense_sweep.jsonl
python3 tools/region_bench.py 2>/dev/null > $OUT/region_step.jsonl
python3 tools/rerank_bench.py 2>/dev/null > $OUT/rerank.jsonl
# launch contract: `bench.py --gpus 2` plainly (starts its own ranks) and under the launcher, ranks sharing the one GPU (gloo)
EVLM_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null > $OUT/bench_dryrun_ranks_on_one_gpu.jsonl
echo "rc $?" >> $OUT/bench_dryrun_ranks_on_one_gpu.jsonl
EVLM_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null >> $OUT/bench_dryrun_ranks_on_one_gpu.jsonl
echo "rc $?" >> $OUT/bench_dryrun_ranks_on_one_gpu.jsonl
# the N > 1 path of the GD step on one GPU with the wire simulated
EVLM_FORCE_REDUCE=1 python3 tools/dp_path_probe.py --reps 1 --only joint,cuts_all,cuts_all_sim,cuts_all_noex,late,late_sim,eager 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/dp_path_probe.txt
# soak: joint graph 3 000 steps; N > 1 segments (one-rank RCCL) 1 500 steps of GD and of the ITR pruning step
( python3 bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-oracle-check --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GD joint graph, 3000 steps:', d['ms_per_step'], 'ms/step', d['last_losses'])"
  python3 bench.py --dropout 0.1 --steps 1500 --warmup 5 --no-cpu-baseline --no-oracle-check --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GD joint graph under dropout 0.1, 1500 steps:', d['ms_per_step'], 'ms/step', d['last_losses'])"
  EVLM_FORCE_REDUCE=1 python3 bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-oracle-check --no-roofline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('GD segments (one-rank RCCL), 1500 steps:', d['ms_per_step'], 'ms/step', d['last_losses'], d['config']['launch'])"
  EVLM_FORCE_REDUCE=1 python3 tools/itr_bench.py 384 64 1500 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ITR-384 segments (one-rank RCCL), 1500 steps:', d)"
  python3 tools/vqa_bench.py 480 32 1000 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('VQA-480 captured step, 1000 steps:', d)" ) > $OUT/soak.txt 2>&1
ls -la $OUT
