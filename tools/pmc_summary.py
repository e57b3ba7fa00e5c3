#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:60]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in sorted(acc.items()):
    if not k.startswith(('void pm::', 'pm::')):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}')
