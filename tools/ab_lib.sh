#!/bin/bash
# usage (GPU box, repo root): tools/ab_lib.sh <other-build.so>   -- same-box A/B of TWO BUILDS of the library (the in-tree one vs another .so, e.g. the
# previous commit's, copied under deepqlearning.jl_amd/build/ so that it travels): config-2 bench three times each, alternating, then config 5 once each.
# Boxes differ by a few per cent; only same-box comparisons decide (this is how the r04 fragment-read regrouping of the dW body was found to LOSE 0.7 %).
other=$1
for i in 1 2 3; do
for so in "" "$other"; do
  DQN_MI355X_LIB=${so:+$PWD/$so} python bench.py --no-cpu-baseline --sustained-seconds 2 --per-call-steps 0 --no-secondary 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|${so:-in-tree} |"
done; done
for so in "" "$other"; do
  DQN_MI355X_LIB=${so:+$PWD/$so} python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|cfg5 ${so:-in-tree} |"
done
