"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-step HBM traffic of the conv kernels.
gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM); units are KiB."""
import csv, glob, json, sys, collections
root, batch, passes_per_run = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
DT = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] else "fp16"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
sym = collections.defaultdict(lambda: collections.defaultdict(float))  # per kernel symbol (first 70 characters)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        fam = "head" if "detect_head" in r["Kernel_Name"] else "conv" if ("conv" in r["Kernel_Name"] or "c2f_fused" in r["Kernel_Name"] or "stem2_fused" in r["Kernel_Name"]) else ("nms" if "nms" in r["Kernel_Name"] else ("decode" if "decode" in r["Kernel_Name"] else
              ("layout" if ("nchw" in r["Kernel_Name"] or "sppf" in r["Kernel_Name"] or "copy_chunks" in r["Kernel_Name"]) else "other")))
        acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        sym[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        sym[r["Kernel_Name"][:70]]["n_" + r["Counter_Name"]] += 1
out = {"round": int(sys.argv[4]) if len(sys.argv) > 4 else 4, "dtype": DT, "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --streams 1 --bare --no-graph" + ("" if DT == "fp16" else " --dtype " + DT),
       "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); counters are KiB; "
                     "%d passes profiled (1 record + 1 warm-up + 2 steps + 5 event-timing passes of the conv launches)" % passes_per_run,
       "batch": batch, "families": {}}
for fam, d in acc.items():
    e = {"fetch_bytes_per_step": 2 * d.get("FETCH_SIZE", 0) * 1024 / passes_per_run, "write_bytes_per_step": d.get("WRITE_SIZE", 0) * 1024 / passes_per_run}
    e["hbm_bytes_per_step"] = e["fetch_bytes_per_step"] + e["write_bytes_per_step"]
    out["families"][fam] = e
out["kernels"] = {k: {"launches_per_step": round(d.get("n_FETCH_SIZE", 0) / passes_per_run, 2), "fetch_GB_per_step": round(2 * d.get("FETCH_SIZE", 0) * 1024 / passes_per_run / 1e9, 3),
                      "write_GB_per_step": round(d.get("WRITE_SIZE", 0) * 1024 / passes_per_run / 1e9, 3)}
                  for k, d in sorted(sym.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))) if 2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0) > 1e4}
print(json.dumps(out, indent=1))
