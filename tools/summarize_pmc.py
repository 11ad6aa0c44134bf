"""tools/summarize_pmc.py <out.json> <gpurun_out/pmc_dir> [...] -- per-kernel averages of every counter found in the
rocprofv3 --pmc passes under the given directories (one pass per directory, tools/pmc_probe.sh), merged into one JSON."""
import collections
import csv
import glob
import json
import os
import sys

out, dirs = sys.argv[1], sys.argv[2:]
table = collections.defaultdict(dict)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not (k.startswith(("void k_", "k_")) or "::k_" in k):
                continue
            agg[(k[:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            table[k][c] = sum(v) / len(v)
            table[k]["launches"] = len(v)
json.dump(table, open(out, "w"), indent=1, sort_keys=True)
for k, d in sorted(table.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:16.1f}")
