#!/bin/bash
# usage: bash tools/pmc_conv2.sh "<shape>" <halo 0|1> <tag> [batch 256] [dtype bf16]   (GPU box)  -> gpurun_out/pmc_<tag>/summary.txt
# SQ counters of one conv kernel incl. VALU / SALU activity (pmc_conv.sh has the DMA / wait view).  DYOLO_BENCH_LIB=<libdyolo.so of another build>
# profiles that build (the ablate build reads the DYOLO_* probes exported around this script).
SHAPE=${1:-128,128,3,1,40}; HALO=${2:-0}; TAG=${3:-conv}; B=${4:-256}; DT=${5:-bf16}
R=$PWD; O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $R/tools/bench_conv.py --batch $B --dtype $DT --halo $HALO --iters 3 ${DYOLO_BENCH_LIB:+--lib $DYOLO_BENCH_LIB} $SHAPE > $O/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 $R/tools/pmc_summary.py $O conv > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete
