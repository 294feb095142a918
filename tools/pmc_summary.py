"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter values over dispatches."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        v2 = v[1:] if len(v) > 1 else v
        print(f"    {c:<28s} {sum(v2) / len(v2):16.1f}   (n={len(v)})")
