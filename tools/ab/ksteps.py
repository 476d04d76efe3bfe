"""Per-launch durations of one kernel (substring) in dispatch order, from a rocprofv3 results .db."""
import glob, sqlite3, sys
path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
pat = sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # only the last N launches
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select d.start, (d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' order by d.start"))
if last:
    rows = rows[-last:]
print(" ".join(f"{d:.0f}" for _, d in rows))
