#!/bin/bash
# usage: bash tools/ab_train.sh <reps> <libA> <libB> ...  (GPU box): alternating training benches (B = 64), ms per step
reps=$1; shift
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    v=$(python tools/bench_with_lib.py $lib --mode train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])")
    echo "$rep $lib $v"
  done
done
