"""VALU / SALU / LDS instruction counts of one kernel attributed to source lines (static; needs an asm built with
-gline-tables-only -save-temps).   python scripts/isa_lines.py <file.s> <mangled-name-prefix> [min_count]"""
import re, sys, collections
path, prefix = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = None
cnt = collections.defaultdict(lambda: [0, 0, 0])
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if t.startswith("v_"): cnt[cur][0] += 1
    elif t.startswith("s_"): cnt[cur][1] += 1
    elif t.startswith("ds_"): cnt[cur][2] += 1
print("totals VALU %d SALU %d LDS %d" % tuple(sum(v[i] for v in cnt.values()) for i in range(3)))
for k, v in sorted(cnt.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
    if v[0] + v[1] >= minc:
        print("%-22s:%-5d VALU %4d SALU %4d LDS %3d" % (k[0], k[1], *v))
