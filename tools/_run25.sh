mkdir -p gpurun_out/r05z
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05z/dp_tgraph.txt; : > $O
for rep in 1 2; do
for v in "EVLM_SEG_TEACHER=fork" "EVLM_SEG_TEACHER=graph"; do
  echo "== rep $rep [$v]" >> $O
  env $v EVLM_FORCE_REDUCE=1 timeout 500 python3 tools/dp_path_probe.py --reps 1 --only cuts_all,cuts_all_sim 2>&1 | grep "cuts_all\|Error" | cut -c1-300 >> $O
done; done
cat $O
