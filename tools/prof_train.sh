#!/bin/bash
# GPU box: kernel stats of the training bench -> gpurun_out/prof_train/
R=$PWD; O=$R/gpurun_out/prof_train${2:+_$2}; rm -rf $O; mkdir -p $O   # $2: a tag (environment such as DYOLO_BN_BEHIND=0 is exported by the caller)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --mode train --batch ${1:-64} --steps 3 --warmup 1 > $O/bench.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/stats -name "*kernel_trace.csv" -delete
tail -2 $O/bench.log
