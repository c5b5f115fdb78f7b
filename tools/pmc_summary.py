"""Mean of one rocprofv3 --pmc counter per kernel name:  python tools/pmc_summary.py <counter_collection.csv> [name filter ...]"""
import collections
import csv
import json
import re
import sys

rows = csv.DictReader(open(sys.argv[1]))
filters = sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in agg.items():
    if filters and not any(f in k for f in filters):
        continue
    name = re.sub(r'\(.*', '', k).replace('void ', '')
    out[name] = {c: {'mean': sum(v) / len(v), 'dispatches': len(v)} for c, v in cs.items()}
print(json.dumps(out, indent=1))
