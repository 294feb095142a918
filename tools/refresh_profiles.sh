#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh TAG'): kernel stats, HBM traffic counters and the
# per-layer table for the default bench, written under gpurun_out/prof_TAG/ (copy what is judged into profiles/).
set -e
TAG=${1:-r02}
B=${2:-256}
DT=${3:-}          # storage dtype of the profiled pass (default: bench.py's own, fp16); e.g. f16x2 -> gpurun_out/prof_TAG_f16x2/
R=$PWD
O=$R/gpurun_out/prof_$TAG${DT:+_$DT}
DTF=${DT:+--dtype $DT}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w && cp -r $R /tmp/w && cd /tmp/w
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --batch $B --steps 30 --warmup 5 --streams 1 --bare $DTF > $O/bench_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --batch $B --steps 2 --warmup 1 --streams 1 --bare --no-graph $DTF > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --batch $B --steps 2 --warmup 1 --streams 1 --bare --no-graph $DTF > $O/bench_write.log 2>&1
mkdir -p $O/pmc && cp -r $O/fetch $O/pmc/ && cp -r $O/write $O/pmc/
python3 tools/traffic_summary.py $O/pmc $B 9 ${TAG//[!0-9]/} "$DT" > $O/traffic.json
python3 bench.py --batch $B --layers $O/layers_b$B.txt --bare $DTF > $O/bench_layers.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
# raw counter csvs are large; keep only the summaries
rm -rf $O/fetch $O/write $O/pmc
find $O/stats -name "*kernel_trace.csv" -delete
echo done
