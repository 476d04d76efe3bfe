"""Per block step of the last sweep in a rocprofv3 .db: period between consecutive diag_kernel starts, and the durations
of the kernels started within it (name initial: d diag, p split, r reduce, s solve, w row, g gw)."""
import glob, sqlite3, sys, re
path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
allk = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
starts = [i for i, r in enumerate(allk) if "leaf_walk" in r[2]]
rows = allk[starts[-1]:]
tag = lambda n: "d" if "diag_kernel" in n else "p" if "panel_split" in n else "r" if "panel_reduce" in n else "s" if "solve_kernel" in n else "w" if "row_kernel" in n else "g" if "gw_kernel" in n else None
dg = [i for i, r in enumerate(rows) if "diag_kernel" in r[2]]
tot = 0.0
for n, (a, b) in enumerate(zip(dg, dg[1:] + [len(rows)])):
    per = ((rows[b][0] if b < len(rows) else max(r[1] for r in rows)) - rows[a][0]) / 1e3
    tot += per
    if n % every == 0:
        ks_ = " ".join(f"{tag(r[2])}{(r[1]-r[0])/1e3:.0f}" for r in rows[a:b] if tag(r[2]))
        print(f"step {n:3d} period {per:7.1f} us  cum {tot/1e3:7.2f} ms   {ks_}")
print(f"total {tot/1e3:.2f} ms over {len(dg)} steps")
