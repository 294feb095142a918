#!/bin/bash
# usage: bash tools/ab_bench.sh <reps> <libA> <libB> [libC ...]   (GPU box) — alternating whole-pass benches (bench.py --bare), img/s per run.
# Box clocks drift between consecutive processes: compare medians of alternating runs, never single pairs.
reps=$1; shift
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    v=$(python tools/bench_with_lib.py $lib --bare --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])")
    echo "$rep $lib $v"
  done
done
