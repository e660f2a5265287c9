"""Spill traffic of one kernel by loop depth: python scripts/spill_depth.py <file.s> <mangled-prefix>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l)
end = next(i for i in range(start, len(lines)) if "; Occupancy" in lines[i])
body = lines[start:end]
wl = collections.Counter(re.findall(r"v_writelane_b32 (v\d+)", "\n".join(body)))
spill = {v for v, c in wl.items() if c >= 4}
depth = 0
cnt = collections.Counter()
for l in body:
    m = re.search(r"Depth=(\d+)", l)
    if l.startswith(".LBB"):
        depth = int(m.group(1)) if m else 0
    elif m and "Loop Header" in l:
        depth = int(m.group(1))
    t = l.strip()
    if t.startswith("v_readlane_b32") and any(re.search(r"\b%s\b" % v, t) for v in spill): cnt[("sgpr-reload", depth)] += 1
    if t.startswith("v_writelane_b32") and any(re.search(r"\b%s\b" % v, t) for v in spill): cnt[("sgpr-spill", depth)] += 1
    if t.startswith("scratch_load"): cnt[("scratch-load", depth)] += 1
    if t.startswith("scratch_store"): cnt[("scratch-store", depth)] += 1
    if t.startswith("v_"): cnt[("VALU", depth)] += 1
    if t.startswith("s_"): cnt[("SALU", depth)] += 1
print("spill VGPRs:", sorted(spill))
for k in sorted(cnt): print(k, cnt[k])
