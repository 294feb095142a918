#!/bin/bash
# GPU box: weight-gradient kernels (product build); second block: the 3x3 kernel without its epilogue (ablate build, DYOLO_WGRAD3_DBG=32)
L=drone-yolo_amd/lib_ablate/libdyolo.so
SH="8,32,3,2,640 32,64,3,2,320 64,64,3,1,160 32,32,3,1,160 64,128,3,2,160 128,256,3,2,80 256,512,3,2,40 64,64,3,2,160 128,128,3,2,80 64,64,3,1,80 128,128,3,1,40 256,256,3,1,20 128,64,3,1,80 64,128,3,1,160"
python tools/bench_wgrad.py $SH
DYOLO_WGRAD3_DBG=32 python tools/bench_wgrad.py --lib $L $SH
