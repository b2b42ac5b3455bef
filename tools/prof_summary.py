"""Per-kernel totals of a rocprofv3 --kernel-trace run (rocpd sqlite output): python tools/prof_summary.py <results.db> [divide_by]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f'total {tot / div:.0f} us over {sum(r[1] for r in rows) / div:.0f} launches (per 1/{div:g})')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f'{r[0][:100]:100s} {r[1] / div:8.1f} {r[2] / div:10.0f} us {r[3]:8.1f} us avg  {100 * r[2] / tot:5.1f} %')
