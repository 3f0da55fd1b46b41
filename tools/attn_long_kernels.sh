#!/bin/bash
# Per-kernel durations of one long-sequence ViT self-attention layer, forward + backward (64 x 12 x 577 and 32 x 12 x 901),
# under rocprofv3 --kernel-trace, for the default library and any others given (A/B of kernel changes):
#   tools/attn_long_kernels.sh [other.so ...]        extra probe arguments through PROBE_ARGS="--kd"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/attnexp; mkdir -p $O
for lib in efficientvlm_amd/libevlm_hip.so "$@"; do
  echo "== $lib ${PROBE_ARGS:-}"
  export EVLM_LIB=$PWD/$lib
  for shape in "64 577" "32 901"; do
    rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 tools/attn_long_probe.py $shape 6 ${PROBE_ARGS:-} > $O/kt.log 2>&1
    python3 tools/rocpd_stats.py $O/kt/kt_results.db 1 $O/kt.csv 5 2>&1 | grep attn_
    rm -rf $O/kt
  done
done
