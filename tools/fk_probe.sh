#!/bin/bash
# GPU box, ablate build WITH the in-kernel probes (make ABLATE=1 KDBG=1 OUT=../lib_ablate): where does the flat-K kernel's time go?  DYOLO_FK_DBG = 0 full, 1 no MFMAs, 2 no LDS-DMA after step 0, 3 no fragment reads, 4 no stores
O=${1:-gpurun_out/fkprobe}
mkdir -p $O
S="160,160,3,1,192 320,320,3,1,96 80,80,3,1,384 2560,640,1,1,96 400,160,1,1,384"
for dt in fp16 fp8; do
  for v in 0 1 2 3 4; do
    DYOLO_FK_DBG=$v python tools/bench_conv.py --lib drone-yolo_amd/lib_ablate/libdyolo.so --dtype $dt --batch 8 --halo 0 $S 2>&1 | grep -v amdgpu.ids | sed "s/dbg=0/fk_dbg=$v/" >> $O/probe_$dt.txt
  done
done
