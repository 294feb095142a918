#!/bin/bash
# GPU box: the x-scale layer shapes (BASELINE config 5, 1536x1536, batch 8) on the flat-K kernel against what ran them before.
# usage: bash tools/fk_bench.sh OUTDIR
O=${1:-gpurun_out/fk}
mkdir -p $O
S3="80,80,3,1,384 160,160,3,1,192 160,160,3,1,384 320,320,3,1,96 320,320,3,1,48 160,64,3,1,384 80,160,3,2,768 160,320,3,2,384 320,640,3,2,192 640,640,3,2,96"
S1="400,160,1,1,384 800,320,1,1,192 1600,640,1,1,96 1280,320,1,1,192 2560,640,1,1,96 480,160,1,1,384 960,320,1,1,192 160,160,1,1,384"
for dt in fp16 fp8; do
  python tools/bench_conv.py --dtype $dt --batch 8 --halo 0 $S3 $S1 > $O/fk_${dt}.txt 2>&1
done
DYOLO_FLAT_K_3X3=0 python tools/bench_conv.py --dtype fp16 --batch 8 --halo 1 $S3 > $O/halo_fp16.txt 2>&1
