"""Summarise rocprofv3 counter_collection CSVs under a directory: mean per dispatch of each counter, per kernel."""
import collections, csv, glob, json, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"] or "k_ss" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
out = collections.defaultdict(dict)
for (k, c), v in sorted(agg.items()):
    out[k][c] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(out, indent=1))
