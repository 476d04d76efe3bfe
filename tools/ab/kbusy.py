"""Wall time vs union of the intervals in which kernels matching a substring run (rocprofv3 results .db)."""
import glob, sqlite3, sys
path = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
pat = sys.argv[2]
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
allk = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# last sweep: from the last leaf_walk kernel on
starts = [i for i, r in enumerate(allk) if "leaf_walk" in r[2]]
rows = allk[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
sel = sorted((r[0], r[1]) for r in rows if pat in r[2])
busy, cur_s, cur_e = 0, None, None
for s, e in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) if cur_e else 0
print(f"sweep wall {(t1-t0)/1e6:.3f} ms; '{pat}' running {busy/1e6:.3f} ms ({100*busy/(t1-t0):.1f} %); sum of its durations {sum(e-s for s,e in sel)/1e6:.3f} ms; launches {len(sel)}")
