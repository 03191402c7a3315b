"""FETCH_SIZE (KiB, rocprofv3 --pmc) per kernel of tools/fetch_calib.hip against the bytes each kernel reads: the factor to apply
usage: python tools/fetch_calib_report.py <counter_collection.csv> <stdout of fetch_calib>"""
import csv, sys, collections
known = {}
for ln in open(sys.argv[2]):
    p = ln.split()
    if len(p) == 3 and p[0] == 'bytes_read':
        known[p[1]] = float(p[2])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE':
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
print(f'{"kernel":18s} {"bytes read":>14s} {"FETCH_SIZE x 1024":>18s} {"bytes / counter":>16s}')
for k, v in sorted(acc.items()):
    if k in known:
        m = 1024.0 * sum(v[1:]) / max(len(v) - 1, 1)          # first launch of each kernel: cold
        print(f'{k:18s} {known[k]:14.0f} {m:18.0f} {known[k] / m:16.3f}')
