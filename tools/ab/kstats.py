"""Print per-kernel average durations from a rocprofv3 results .db (kernel names containing any given substring)."""
import glob
import sqlite3
import sys

path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
pats = sys.argv[2:]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"
for name, n, avg in c.execute(q):
    if not pats or any(p in name for p in pats):
        short = name.split("_GLOBAL__N_1")[-1][:40]
        print(f"   {short:40s} n={n:5d} avg={avg:9.2f} us  total={n * avg / 1e3:9.3f} ms")
