mkdir -p gpurun_out/r05z
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
EVLM_FORCE_REDUCE=1 timeout 900 python3 tools/dp_path_probe.py --reps 1 --only joint,cuts_all,cuts_all_sim,nosplit_sim,cuts_v4_sim,cuts_v3_sim,cuts_v2_sim,cuts_421_sim,cuts_vit_sim,cuts_none_sim,bf16 2>&1 | grep "joint\|cuts_\|nosplit\|bf16\|Error" | cut -c1-260 > gpurun_out/r05z/dp_cuts.txt
cat gpurun_out/r05z/dp_cuts.txt
