"""Per-kernel breakdown of ONE forward from a rocprofv3 kernel-trace database (B = 1 latency analysis)."""
import sqlite3, re, collections, sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select d.start,d.end,s.kernel_name,d.grid_size_x,d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'time_mlp' in r[2]]
fw = rows[idx[-2]:idx[-1]]
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '')
tot = sum(r[1] - r[0] for r in fw) / 1e3
print("launches", len(fw), "sum us %.1f" % tot, "span us %.1f" % ((fw[-1][1] - fw[0][0]) / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in fw:
    agg[short(r[2])][0] += 1; agg[short(r[2])][1] += (r[1] - r[0]) / 1e3
for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%8.1f us %4d  %s" % (t, k, n[:100]))
print("top launches:")
for r in sorted(fw, key=lambda r: -(r[1] - r[0]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%7.1f us grid %5d  %s" % ((r[1] - r[0]) / 1e3, r[3] // r[4], short(r[2])[:80]))
