"""mean duration of every kernel of the MultiWalker step in steady state (second half of a rocprofv3 --kernel-trace run of
scripts/mw_steady.py): python scripts/mw_kernels.py <results.db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x from %s d join %s s on d.kernel_id = s.id where s.kernel_name like '%%mw_%%' order by d.start" % (kd, sym)))
rows = rows[len(rows) // 2:]
acc = collections.defaultdict(list)
for name, st, en, gx in rows:
    acc[(name.split('(')[0][-40:], gx)].append((en - st) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print("%-44s grid %8d  n %5d  mean %8.1f us  p50 %8.1f  max %8.1f" % (k[0], k[1], len(v), sum(v) / len(v), v[len(v) // 2], v[-1]))
