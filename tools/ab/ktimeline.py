"""Timeline (start offset, duration, name) of the kernels of the last sweep in a rocprofv3 .db, for steps [a, b)."""
import glob, sqlite3, sys
path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
skip, count = int(sys.argv[2]), int(sys.argv[3])
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
allk = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
starts = [i for i, r in enumerate(allk) if "leaf_walk" in r[2]]
rows = allk[starts[-1]:][skip:skip + count]
t0 = rows[0][0]
import re
for s, e, n in rows:
    m = re.search(r"(\w+_kernel)", n)
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f}  end {(e-t0)/1e3:9.1f}  {m.group(1) if m else n[:30]}")
