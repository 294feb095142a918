#!/bin/bash
# usage: bash tools/ab_bench_env.sh <reps> "<ENV lib>" ...   each arm = "VAR=val,... path/to/libdyolo.so" (GPU box); alternating whole-pass benches
reps=$1; shift
for rep in $(seq 1 $reps); do
  for arm in "$@"; do
    envs=${arm% *}; lib=${arm##* }
    v=$(env ${envs//,/ } python tools/bench_with_lib.py $lib --bare --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'])")
    echo "$rep [$envs] $lib $v"
  done
done
