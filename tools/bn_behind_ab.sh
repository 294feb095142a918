#!/bin/bash
# GPU box: bash tools/bn_behind_ab.sh OUT REPS — training-step A/B of the BatchNorm-backward sums from the input-gradient epilogue
# (DYOLO_BN_BEHIND=0: off; ablate build: DYOLO_BNB_WGS=2 compiles the 64-channel form for two workgroups per CU, no spills)
O=${1:-gpurun_out/bnb}; REPS=${2:-2}
mkdir -p $O
for rep in $(seq 1 $REPS); do
  for arm in "DYOLO_BN_BEHIND=0" "DYOLO_BN_BEHIND=1" "DYOLO_BN_BEHIND=1 DYOLO_BNB_WGS=2"; do
    env $arm python tools/bench_with_lib.py drone-yolo_amd/lib_ablate/libdyolo.so --mode train --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep [$arm]', d['ms_per_step'], d.get('gpu_ms_per_step'))" | tee -a $O/ab.txt
  done
done
