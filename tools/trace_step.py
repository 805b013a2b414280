"""Print the kernel timeline of the last two steps from a rocprofv3 rocpd database (default output format):
    rocprofv3 --kernel-trace -d gpurun_out/prof -o run -- python3 <program>;  python tools/trace_step.py gpurun_out/prof/run_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in cur.execute("select name, total_calls, average from top_kernels limit 10"):
    print(f"{r[0][:64]:64s} calls {r[1]:6d}  avg {r[2]:9.2f} us")
rows = list(cur.execute("select name,start,end from kernels order by start"))
rows = [r for r in rows if 'gpf::' in r[0]][-n:]
base = rows[0][1]
for name, s, e in rows:
    print(f"{(s - base) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {name[:60]}")
