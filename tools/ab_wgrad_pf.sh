#!/bin/bash
# GPU box: weight-gradient kernels, product build against the per-tap kernel with atomics for 1x1 layers (ablate build, DYOLO_NO_WGRAD1)
L=drone-yolo_amd/lib_ablate/libdyolo.so
SH="64,64,1,1,160 96,64,1,1,160 128,128,1,1,80 192,128,1,1,80 256,128,1,1,80 384,256,1,1,40 256,256,1,1,40 512,512,1,1,20 768,512,1,1,20 1024,512,1,1,20 384,128,1,1,80"
python tools/bench_wgrad.py $SH
DYOLO_NO_WGRAD1=1 python tools/bench_wgrad.py --lib $L $SH
