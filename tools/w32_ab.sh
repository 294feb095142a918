#!/bin/bash
# GPU box: bash tools/w32_ab.sh OUTDIR — timing A/B and SQ counters of the 128 x 128 flat-K tile on 16x16x32 and on 32x32x16 MFMAs (ablate build)
O=${1:-gpurun_out/w32}
mkdir -p $O
python tools/w32_ab.py --dtype fp16 --batch 8 > $O/ab_fp16_b8.txt 2>&1 || exit 1
python tools/w32_ab.py --dtype bf16 --batch 8 256,256,3,1,80 768,256,1,1,80 > $O/ab_bf16_b8.txt 2>&1 || exit 1
export DYOLO_BENCH_LIB=$PWD/drone-yolo_amd/lib_ablate/libdyolo.so DYOLO_FK_BN=128 DYOLO_NO_VGEMM=1 DYOLO_NO_GLDS=1
for v in 0 1; do
  export DYOLO_FK_W32=$v
  bash tools/pmc_conv2.sh 256,256,3,1,80 0 w32_${v}_3x3 8 fp16 > /dev/null 2>&1
  bash tools/pmc_conv2.sh 768,256,1,1,80 0 w32_${v}_1x1 8 fp16 > /dev/null 2>&1
  cp gpurun_out/pmc_w32_${v}_3x3/summary.txt $O/pmc_w32_${v}_3x3.txt
  cp gpurun_out/pmc_w32_${v}_1x1/summary.txt $O/pmc_w32_${v}_1x1.txt
done
cat $O/ab_fp16_b8.txt $O/ab_bf16_b8.txt
