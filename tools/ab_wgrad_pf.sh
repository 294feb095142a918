#!/bin/bash
# GPU box: conv_wgrad3x3 A/B: product build, and the ablate build in drone-yolo_amd/lib_ablate with probes
L=drone-yolo_amd/lib_ablate/libdyolo.so
SH="32,64,3,2,320 64,64,3,1,160 32,32,3,1,160 64,128,3,2,160 128,256,3,2,80 256,512,3,2,40 64,64,3,2,160 128,128,3,2,80 64,64,3,1,80 128,128,3,1,40 256,256,3,1,20 128,64,3,1,80"
python tools/bench_wgrad.py $SH
echo "atomics instead of the workspace:"
DYOLO_WGRAD3_ATOMICS=1 python tools/bench_wgrad.py --lib $L $SH
