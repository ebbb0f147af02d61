"""Mean per-dispatch value of every counter in a rocprofv3 `*_counter_collection.csv`, grouped by kernel."""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name") or row.get("kernel_name") or "?"
        c = row.get("Counter_Name") or row.get("counter_name")
        v = row.get("Counter_Value") or row.get("counter_value")
        if c is None or v is None:
            continue
        acc[k[:90]][c].append(float(v))
out = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
