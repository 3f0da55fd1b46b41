#!/bin/bash
# A/B of the GD step on ONE box (devices of the pool differ by several per cent): alternates the variants given as
# "ENV=VALUE" strings ("-" = defaults) and prints ms/step of each run.   tools/ab_step.sh 3 - EVLM_NO_FORK=1
reps=$1; shift
for r in $(seq $reps); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    ms=$(env $e python bench.py --no-cpu-baseline --no-oracle-check --no-roofline --steps 20 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $r  ${v}  ${ms} ms/step"
  done
done
