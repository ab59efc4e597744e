#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel: sum and per-launch mean of every counter."""
import csv, glob, sys, collections, json
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
        for k in acc:
            out.setdefault(k, {}).update({c: {'sum': v, 'per_launch': v / max(1, len(n[k]))} for c, v in acc[k].items()})
            out[k]['launches'] = len(n[k])
print(json.dumps(out, indent=1))
