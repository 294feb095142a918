"""Summarise a rocprofv3 --kernel-trace CSV: average duration per (kernel, grid size)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=lambda p: -__import__("os").path.getmtime(p))[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"][:70], r["Grid_Size_X"], r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc.setdefault(key, []).append(d)
for k, v in acc.items():
    v2 = v[1:] if len(v) > 1 else v
    print(f"{k[0]:<70s} grid={k[1]:>8s} lds={k[2]:>7s} vgpr={k[3]:>4s} n={len(v):3d} avg_us={sum(v2)/len(v2):8.1f} min={min(v2):8.1f}")
