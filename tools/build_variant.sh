#!/bin/bash
# usage (container, repo root): tools/build_variant.sh <name> [-DDEF ...]   -- a second build of the library with extra compile-time defines, linked as
# deepqlearning.jl_amd/build/<name>.so (git-ignored, travels to the GPU box).  Select it with DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/<name>.so;
# tools/run_ab.sh alternates it with the in-tree build on one box.  `tools/build_variant.sh base` with no defines snapshots the current sources (the "before" of an A/B).
set -e
name=$1; shift
P=deepqlearning.jl_amd; O=$P/build/v_$name; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -w $*"
pids=()
for f in $P/csrc/*.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $O/$(basename ${f%.hip}).o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/$name.so $O/*.o -ldl
rm -rf $O
echo built $P/build/$name.so
