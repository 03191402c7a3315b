import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r["Kernel_Name"].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   ", c.ljust(24), f"{x:.3e}")
