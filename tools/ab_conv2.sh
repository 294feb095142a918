#!/bin/bash
# usage: bash tools/ab_conv2.sh shape...   (GPU box): --halo 0 (row layout: LDS-DMA GEMM) against --halo 1 (default layout rule), alternating
for rep in 1 2 3; do
  echo "== rows rep $rep"; python tools/bench_conv.py --halo 0 --batch 256 --iters 30 "$@" | awk '{print $1, $5, $6}'
  echo "== default rep $rep"; python tools/bench_conv.py --halo 1 --batch 256 --iters 30 "$@" | awk '{print $1, $5, $6}'
done
