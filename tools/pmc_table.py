"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files:  python tools/pmc_table.py <dir-or-csv> [name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

paths = []
for a in sys.argv[1:]:
    if os.path.isdir(a):
        paths += glob.glob(os.path.join(a, '**', '*counter_collection.csv'), recursive=True)
    elif a.endswith('.csv'):
        paths.append(a)
filters = [a for a in sys.argv[1:] if not os.path.exists(a)]
acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(lambda: defaultdict(set))
for p in paths:
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name'].split('(')[0][:60]
        if filters and not any(f in k for f in filters):
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print(f'    {c:32s} {acc[k][c] / max(len(n[k][c]), 1):16.1f}   ({len(n[k][c])} launches)')
