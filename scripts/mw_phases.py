"""per-launch durations of the MultiWalker phase kernels in steady state, from a rocprofv3 --kernel-trace database
    rocprofv3 --kernel-trace -d gpurun_out/mwprof -o mw -- python scripts/mw_steady.py --quick ; python scripts/mw_phases.py gpurun_out/mwprof/mw_results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id where s.kernel_name like '%%mw_phase%%' order by d.start" % (kd, sym)))
seq = [(int(re.search(r'ILi(\d)E', r[0]).group(1)), (r[2] - r[1]) / 1e3) for r in rows]
# a step() call = [RESET (spares)] COLLIDE SOLVE TOI [RESET COLLIDE SOLVE TOI (pass 1)]: cut the launch sequence at every TOI that is
# followed by a launch that can only begin a new call (the spares' RESET, or COLLIDE right after a pass-1 TOI)
names = {0: "reset", 1: "collide", 2: "solve", 3: "toi"}
pattern = None
steps, cur_step = [], []
full = [p for p, _ in seq]
# find the period of the sequence in its steady part
for per in (8, 7, 4, 3):
    mid = len(full) // 2
    if full[mid:mid + per] == full[mid + per:mid + 2 * per] and len(set(full[mid:mid + per])) > 1: pattern = per; break
if pattern is None: raise SystemExit("no periodic launch pattern found")
# align to a period start: the first index >= 100 * period where the period begins with what the steady part begins a call with
start = next(i for i in range(len(full) // 3, len(full)) if full[i:i + pattern] == full[i + pattern:i + 2 * pattern] and (full[i] == 0 if pattern in (8, 4) else full[i] == 1) and (pattern != 8 or full[i + 1] == 1))
if pattern == 8:   # RESET COLLIDE SOLVE TOI twice: the first group is the one that does the work
    w = lambda o: sum(seq[i][1] for i in range(start + o + 1, len(seq) - 8, 8))
    if w(4) > w(0): start += 4
calls = [seq[i:i + pattern] for i in range(start, len(seq) - pattern + 1, pattern)]
sel = calls[10:70]
labels = ["%s%s" % (names[p], "'" if k >= (4 if pattern == 8 else 3) and pattern >= 7 else "") for k, (p, _) in enumerate(calls[0])]
avg = [sum(c[k][1] for c in sel) / len(sel) for k in range(pattern)]
print("%d launches per call (' = pass 1), mean of %d calls: " % (pattern, len(sel)) + "  ".join("%s %.0f" % (l, a) for l, a in zip(labels, avg)) + "  | total %.0f us" % sum(avg))
