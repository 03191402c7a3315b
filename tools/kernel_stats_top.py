import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:16]:
    print(r["Name"][:100].replace("(anonymous namespace)::",""), r["Calls"], round(int(r["TotalDurationNs"])/1e6,2), "ms")
