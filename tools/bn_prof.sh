#!/bin/bash
# GPU box: per-kernel times of tools/bench_bn.py under rocprofv3 for a list of ablate configurations ("VAR=..;VAR=.." per argument)
R=$PWD; L=$R/drone-yolo_amd/lib_ablate/libdyolo.so
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1)); O=$R/gpurun_out/bn_prof/$i; rm -rf $O; mkdir -p $O
  for kv in ${cfg//;/ }; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_bn.py --lib $L > $O/bench.log 2>&1
  for kv in ${cfg//;/ }; do unset ${kv%%=*}; done
  echo "== $cfg"; f=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bn_" in r["Name"]:
        print(f'{r["Name"][:70]:<70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
  find $O -name "*kernel_trace.csv" -delete
done
