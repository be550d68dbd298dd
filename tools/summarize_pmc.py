"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per dispatch, per kernel."""
import csv
import collections
import json
import sys

path, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        acc[k[:80]][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
res = {k: {c: {"mean": sum(v) / len(v), "median": sorted(v)[len(v) // 2], "max": max(v), "n": len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out, "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -max(x["mean"] for x in kv[1].values()))[:8]:
    print(k, d)
