"""per-launch durations of the MultiWalker phase kernels in steady state, from a rocprofv3 --kernel-trace database
    rocprofv3 --kernel-trace -d gpurun_out/mwprof -o mw -- python scripts/mw_steady.py --quick ; python scripts/mw_phases.py gpurun_out/mwprof/mw_results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id where s.kernel_name like '%%mw_phase%%' order by d.start" % (kd, sym)))
seq = [(int(re.search(r'ILi(\d)E', r[0]).group(1)), (r[2] - r[1]) / 1e3) for r in rows]
# a step = COLLIDE SOLVE TOI [RESET COLLIDE SOLVE TOI]; find steps by scanning
names = {0: "reset", 1: "collide", 2: "solve", 3: "toi"}
steps, i = [], 0
while i + 2 < len(seq):
    if [p for p, _ in seq[i:i + 3]] == [1, 2, 3]:
        st = {"collide0": seq[i][1], "solve0": seq[i + 1][1], "toi0": seq[i + 2][1]}
        i += 3
        if i + 3 < len(seq) and [p for p, _ in seq[i:i + 4]] == [0, 1, 2, 3] and not (i + 6 < len(seq) and [p for p, _ in seq[i + 4:i + 7]] != [1, 2, 3] and False):
            st.update({"reset": seq[i][1], "collide1": seq[i + 1][1], "solve1": seq[i + 2][1], "toi1": seq[i + 3][1]})
            i += 4
        steps.append(st)
    else:
        i += 1
n = len(steps)
for lo, hi in ((100, 160), (n // 2 + 100, n // 2 + 160)):
    sel = steps[lo:hi]
    if not sel: continue
    keys = ["collide0", "solve0", "toi0", "reset", "collide1", "solve1", "toi1"]
    avg = {k: sum(s.get(k, 0.0) for s in sel) / len(sel) for k in keys}
    print("steps %d..%d: " % (lo, hi) + "  ".join("%s %.0f" % (k, avg[k]) for k in keys) + "  | total %.0f us" % sum(avg.values()))
