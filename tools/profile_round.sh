#!/bin/bash
# Round profile bundle (run on the GPU box from the repo root): default bench line, rocprofv3 kernel stats of the bench
# command, and the PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) summarised per kernel.  Only small summaries are kept.
#   EVLM_COMMIT=<sha> OUT=gpurun_out/r03p tools/profile_round.sh
set -u
OUT=${OUT:-gpurun_out/r03p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_1gpu.json
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-oracle-check > $OUT/kt.log 2>&1
python3 tools/rocpd_stats.py $OUT/kt/kt_results.db 31 $OUT/kernel_stats.csv 70 > $OUT/kernel_stats.txt 2>&1
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
ls -la $OUT
