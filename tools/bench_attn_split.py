"""Per-kernel times of the attention launches from a rocprofv3 kernel trace of tools/bench_attn.py (diagnostics):
    rocprofv3 --kernel-trace -d <dir> -o r -- python3 tools/bench_attn.py 32 8 planes ; python tools/bench_attn_split.py <dir>/r_results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, n, avg in db.execute("select name, count(*), avg(end-start)/1e3 from kernels where name like '%k_attn%' group by name"):
    print(f"{name[name.index('k_attn'):].split('(')[0]:24s} n={n:4d} avg {avg:8.1f} us")
