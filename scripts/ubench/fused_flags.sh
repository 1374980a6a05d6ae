# the fused shrink pass built with different options (one launch for 45 bands of 4096 x 2732), alternating on one box
cd $GRAFT_REPO_ROOT
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Iart_amd/csrc -mllvm -amdgpu-sched-strategy=max-memory-clause"
hipcc $F scripts/ubench/fused_bench.hip art_amd/csrc/shrinkblur.hip -o /tmp/fb0
hipcc $F ${FS_VARIANT:--DFS_NT_STORE} scripts/ubench/fused_bench.hip art_amd/csrc/shrinkblur.hip -o /tmp/fb1
for i in 1 2 3; do echo "default: $(/tmp/fb0 4096 2732 2 | tail -1)"; echo "variant: $(/tmp/fb1 4096 2732 2 | tail -1)"; done
