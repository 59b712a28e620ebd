#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per dispatch."""
import csv, sys, collections, glob
def short(n):
    n = n.split('(')[0]
    return n.replace('void ', '')[:60]
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        acc[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', path)
    for k, d in acc.items():
        if not (k.startswith('k_') or 'k_' in k[:8]):
            continue
        print('  %-58s' % k, '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
