#!/bin/bash
# usage (GPU box, repo root): tools/run_ab.sh [-n reps] [-c 2|5|both] [-t "pytest -k expr"] [-e "VAR=val ..."] <variant.so | ""> ...
#   same-box A/B of library builds (tools/build_variant.sh) and / or engine switches: the in-tree build and every named variant take turns, `reps` times, on the
#   config-2 bench (steps/s + per-launch table row) and / or the config-5 bench; -t first runs the parity tests selected by the expression against EVERY build
#   (a variant that is not bit-exact is reported and skipped); -e sets environment switches for the VARIANT runs only ("" as the variant = the in-tree build with -e).
# Boxes differ by a few per cent: only same-box, alternating comparisons decide.  One script instead of one launcher per experiment (r05 left 28 of them).
reps=3; cfg=both; kexpr=""; venv=""
while getopts "n:c:t:e:" o; do case $o in n) reps=$OPTARG;; c) cfg=$OPTARG;; t) kexpr=$OPTARG;; e) venv=$OPTARG;; esac; done; shift $((OPTIND - 1))
mkdir -p gpurun_out
builds=("in-tree"); for v in "$@"; do builds+=("$v"); done
libof() { [ "$1" = "in-tree" ] && echo "" || { [ -z "$1" ] && echo "" || echo "$PWD/deepqlearning.jl_amd/build/$1"; }; }
envof() { [ "$1" = "in-tree" ] && echo "" || echo "$venv"; }
ok=()
for b in "${builds[@]}"; do
  if [ -n "$kexpr" ]; then
    r=$(env $(envof "$b") DQN_MI355X_LIB=$(libof "$b") timeout 1200 python -m pytest tests -x -q -m gpu -k "$kexpr" 2>&1 | grep -E "^E  |passed|failed|rror" | tail -3)
    echo "parity [${b:-in-tree+env}] $r"; echo "$r" | grep -q "failed\|rror" && continue
  fi
  ok+=("$b")
done
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
L = d["roofline"]["launches"]
print("%-28s %8.1f steps/s  %7.2f us/step  " % (sys.argv[2], d["value"], 1e3 * d["ms_per_step"]) + "  ".join("%s %.1f" % (x["launch"].split("+")[0], x["avg_us"]) for x in L))
PY
}
for i in $(seq $reps); do for b in "${ok[@]}"; do
  if [ "$cfg" != 5 ]; then env $(envof "$b") DQN_MI355X_LIB=$(libof "$b") timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 2 --per-call-steps 0 --no-secondary --env-steps 0 2>/dev/null > gpurun_out/_ab.json && line gpurun_out/_ab.json "cfg2 ${b:-in-tree+env}"; fi
  if [ "$cfg" != 2 ]; then env $(envof "$b") DQN_MI355X_LIB=$(libof "$b") timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null > gpurun_out/_ab.json && line gpurun_out/_ab.json "cfg5 ${b:-in-tree+env}"; fi
done; done
rm -f gpurun_out/_ab.json
