"""per-kernel count / average / total (us) from a rocprofv3 rocpd database (the default output format of this ROCm)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 4 desc"
for r in cur.execute(q):
    print(f'{r[0][:100]:100s} {r[1]:6d} {r[2]:10.1f} {r[3]:12.1f}')
