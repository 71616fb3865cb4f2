#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of the sensing kernel: mean counter values and kernel times."""
import collections, csv, glob, os, sys
def newest(pattern):
    m = sorted(glob.glob(pattern), key=os.path.getmtime)
    return m[-1:] 

root = sys.argv[1]
for g in sorted(glob.glob(root + "/*/")):
    cc = newest(g + "*/*_counter_collection.csv")
    if not cc:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'sense_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = newest(g + "*/*_kernel_trace.csv")
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(kt[0]))
         if 'sense_kernel' in r['Kernel_Name']]
    tail = d[-5:]
    print(g.rstrip('/').split('/')[-1], "kernel us (last 5): " + " ".join("%.0f" % x for x in tail))
    for k, v in sorted(acc.items()):
        v = v[-5:]
        print("   %-28s %.4g" % (k, sum(v) / len(v)))
