#!/bin/bash
# usage: bash tools/ab_conv.sh "<ENV_A>" "<ENV_B>" shape...   (GPU box)
# Alternates the two environments three times per shape set and prints every run: box clocks drift by several per cent
# between consecutive processes, so a single A/B pair proves nothing.
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  echo "== A ($A) rep $rep"; env $A python tools/bench_conv.py --halo 0 --batch 256 --iters 30 "$@" | awk '{print $1, $5, $6}'
  echo "== B ($B) rep $rep"; env $B python tools/bench_conv.py --halo 0 --batch 256 --iters 30 "$@" | awk '{print $1, $5, $6}'
done
