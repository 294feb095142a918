#!/bin/bash
# usage: bash tools/pmc_c2f.sh [tag]   (GPU box)  -> gpurun_out/pmc_c2f_<tag>/summary.txt: SQ counters of both dy_c2f_fused variants
R=$PWD; O=$R/gpurun_out/pmc_c2f_${1:-r02}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $R/tools/bench_c2f.py --iters 3 > $O/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 $R/tools/pmc_summary.py $O c2f_fused > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete
