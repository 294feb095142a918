#!/bin/bash
# GPU box: which launches of a training step go to the generic kernel / to torch: per (symbol, grid) calls and time of the kernel trace
R=$PWD; O=$R/gpurun_out/prof_train_detail; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --mode train --batch ${1:-64} --steps 2 --warmup 1 > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if not ("igemm" in name or "at::native" in name or "copy_chunks" in name or "rocclr" in name):
        continue
    key = (name[:90], r["Grid_Size_X"], r["Workgroup_Size_X"])
    acc[key][0] += 1
    acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
with open(sys.argv[1] + "/summary.txt", "w") as out:
    for (name, grid, wg), (n, us) in rows[:60]:
        out.write(f"{us:10.1f} us  {n:5d} calls  grid {grid:>9s} wg {wg:>4s}  {name}\n")
PY
find $O/trace -name "*.csv" -delete
head -45 $O/summary.txt
