"""Per-kernel mean of every counter in rocprofv3 counter_collection.csv files.
usage: python tools/pmc_kernel.py KERNEL_SUBSTRING file.csv [file.csv ...]"""
import collections
import csv
import sys

pat = sys.argv[1]
for path in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if pat in r['Kernel_Name']:
            acc[r['Kernel_Name'][:90]][r['Counter_Name']] += float(r['Counter_Value'])
            disp[r['Kernel_Name'][:90]].add(r['Dispatch_Id'])
    for k, d in acc.items():
        n = len(disp[k])
        print(k, 'dispatches', n)
        for c, v in sorted(d.items()):
            print(f'   {c:28s} {v / n:16.0f}')
