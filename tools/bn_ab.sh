#!/bin/bash
# GPU box: A/B of the BatchNorm pass variants (ablate build): rows in flight per thread of the apply kernels, grid cap
L=drone-yolo_amd/lib_ablate/libdyolo.so
for rep in 1 2; do
for cfg in "DYOLO_BN_UNR=1 DYOLO_BN_BUNR=1" "DYOLO_BN_UNR=2 DYOLO_BN_BUNR=2" "DYOLO_BN_UNR=4 DYOLO_BN_BUNR=4" "DYOLO_BN_UNR=4 DYOLO_BN_BUNR=2 DYOLO_BN_GRID=2048" "DYOLO_BN_UNR=4 DYOLO_BN_BUNR=2 DYOLO_BN_RUNR=8" "DYOLO_BN_UNR=2 DYOLO_BN_BUNR=2 DYOLO_BN_GRID=8192"; do
  echo "== $cfg rep $rep"; env $cfg python tools/bench_bn.py --lib $L "$@" | tail -9
done
done
