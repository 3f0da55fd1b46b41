mkdir -p gpurun_out/r05i
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in gd itr vqa; do timeout 600 python3 tools/find_small_ops.py $k 2>&1 | grep -v amdgpu.ids > gpurun_out/r05i/aten_$k.txt; done
head -5 gpurun_out/r05i/aten_*.txt
