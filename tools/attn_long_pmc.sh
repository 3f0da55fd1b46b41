#!/bin/bash
# SQ counter passes over one long-sequence ViT self-attention layer, forward + backward (tools/attn_long_probe.py), per kernel:
# wave cycles and where they go (parked at a wait / issue stalls / issuing), instruction counts, LDS activity and bank conflicts,
# matrix-pipe busy cycles.  Counters only, no other tracing (the pool's rule).   tools/attn_long_pmc.sh [B L] [--kd]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/attnpmc; mkdir -p $O
ARGS="${1:-64} ${2:-577} 3 ${3:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/p1 -o p1 -- python3 tools/attn_long_probe.py $ARGS > $O/p1.log 2>&1
python3 tools/pmc_any.py $O/p1/p1_results.db attn
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/p2 -o p2 -- python3 tools/attn_long_probe.py $ARGS > $O/p2.log 2>&1
python3 tools/pmc_any.py $O/p2/p2_results.db attn
rocprofv3 --kernel-trace --pmc SQ_WAVES GRBM_GUI_ACTIVE -d $O/p3 -o p3 -- python3 tools/attn_long_probe.py $ARGS > $O/p3.log 2>&1
python3 tools/pmc_any.py $O/p3/p3_results.db attn
rm -rf $O/p1 $O/p2 $O/p3
