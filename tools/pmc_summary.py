#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (tools/pmc_profile.sh): mean counter value per launch of
each kernel, summed over the dispatch's dimensions/instances the way rocprofv3 reports them."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))    # kernel -> counter -> dispatch -> value
for path in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
            acc[name][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = {}
for k, ctrs in sorted(acc.items()):
    out[k] = {c: sum(v.values()) / len(v) for c, v in sorted(ctrs.items())}
    out[k]["launches_seen"] = max(len(v) for v in ctrs.values())
json.dump(out, sys.stdout, indent=1)
