set -e
cd $GRAFT_REPO_ROOT
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iart_amd/csrc -mllvm -amdgpu-sched-strategy=max-memory-clause"
hipcc $F -DFS_PROFILE scripts/ubench/fused_bench.hip art_amd/csrc/shrinkblur.hip -o /tmp/fbp
hipcc $F scripts/ubench/fused_bench.hip art_amd/csrc/shrinkblur.hip -o /tmp/fb
for m in 0 1 2; do /tmp/fbp 4096 2732 $m | tail -1; done
for m in 0 1 2; do /tmp/fb 4096 2732 $m | tail -1; done
