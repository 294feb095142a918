#!/bin/bash
# usage: bash tools/pmc_c2f_ta.sh [tag]  (GPU box): texture-addresser / L1 view of dy_c2f_fused -> gpurun_out/pmc_c2f_ta_<tag>/summary.txt
R=$PWD; O=$R/gpurun_out/pmc_c2f_ta_${1:-r02}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $R/tools/bench_c2f.py --iters 3 > $O/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $O/$tag.log; }
done
python3 $R/tools/pmc_summary.py $O c2f_fused > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*counter_collection.csv" -delete
