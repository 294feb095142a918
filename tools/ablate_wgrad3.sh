#!/bin/bash
# GPU box: timing probes of conv_wgrad3x3_kernel (ablate build in drone-yolo_amd/lib_ablate): which part of a step costs what.
# bits of DYOLO_WGRAD3_DBG: 1 no global loads, 32 no epilogue stores / atomics
for sh in ${@:-64,64,3,1,160 128,128,3,1,40 64,128,3,2,160}; do
  for dbg in 0 32 1 33; do
    DYOLO_WGRAD3_DBG=$dbg python tools/bench_wgrad.py --lib drone-yolo_amd/lib_ablate/libdyolo.so $sh
  done
done
