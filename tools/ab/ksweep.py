"""Last sweep of a rocprofv3 kernel-trace .db, block step by block step: for every launch its start offset and duration
(d diag, w row, s solve, L solve_lite, p split, r reduce), plus the union busy time per kernel family.
   python3 tools/ab/ksweep.py <file.db | dir> [first_step] [n_steps]"""
import glob, os, sqlite3, sys
arg = sys.argv[1]
path = arg if os.path.isfile(arg) else glob.glob(arg + "/**/*.db", recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
allk = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
starts = [i for i, r in enumerate(allk) if "leaf_walk" in r[2]]
rows = allk[starts[-1]:]
def tag(n):
    for k, t in (("diag_pre", "e"), ("diag_kernel", "d"), ("panel_split", "p"), ("panel_reduce", "r"), ("solve_lite", "L"), ("solve_", "s"), ("row_kernel", "w"),
                 ("sync_gate", "g"), ("sync_publish", "u")):
        if k in n:
            return t
    return None
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"sweep wall {(t1 - t0) / 1e6:.3f} ms, {len(rows)} launches")
fam = {}
for s, e, n in rows:
    t = tag(n)
    if t:
        fam.setdefault(t, []).append((s, e))
for t, iv in sorted(fam.items()):
    iv.sort()
    busy, cs, ce = 0, None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"  {t}: launches {len(iv):4d}  union {busy / 1e6:8.3f} ms  sum {sum(e - s for s, e in iv) / 1e6:8.3f} ms")
dg = [i for i, r in enumerate(rows) if "diag_kernel" in r[2]]
for n, (a, b) in enumerate(zip(dg, dg[1:] + [len(rows)])):
    if n < first or n >= first + count:
        continue
    items = " ".join(f"{tag(r[2])}@{(r[0] - t0) / 1e3:.0f}+{(r[1] - r[0]) / 1e3:.0f}" for r in rows[a:b] if tag(r[2]))
    print(f"step {n:3d}: {items}")
