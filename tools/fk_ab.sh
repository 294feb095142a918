#!/bin/bash
# GPU box: A/B of the flat-K kernel's tile choices (ablate build): bash tools/fk_ab.sh OUTDIR
O=${1:-gpurun_out/fkab}
mkdir -p $O
S3="80,80,3,1,384 160,160,3,1,192 160,160,3,1,384 320,320,3,1,96 80,160,3,2,768 160,320,3,2,384 320,640,3,2,192 640,640,3,2,96"
S1="400,160,1,1,384 800,320,1,1,192 1600,640,1,1,96 1280,320,1,1,192 2560,640,1,1,96 480,160,1,1,384 160,160,1,1,384"
for dt in fp16 fp8; do
  for t in 0 2; do
    DYOLO_FK_TILE=$t python tools/bench_conv.py --lib drone-yolo_amd/lib_ablate/libdyolo.so --dtype $dt --batch 8 --halo 0 $S3 $S1 > $O/tile${t}_${dt}.txt 2>&1
  done
done
