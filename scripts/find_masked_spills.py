"""Static check of a gfx950 .s file (hipcc -save-temps) for register-allocator spill copies that execute under a NARROWED exec mask:
a `v_accvgpr_write_b32 aN, vM` placed at the head of a control-flow JOIN block, before the `s_or_b64 exec, exec, sK` that re-enables
the lanes the divergent region had masked off, where vM is live INTO that region from the wider mask (no definition of vM between the
`s_and_saveexec_b64 sK` that opened the region and the copy).  The lanes the region masked off never get their copy; a later reload
under the full mask hands them whatever the accumulation register held before.

    python scripts/find_masked_spills.py file.s [kernel-name-substring]

This is the root cause of the fault of mwk_c10::mw_step_kernel<7> (profiles/r05_multiwalker/rocgdb_c10_fused.txt): hipcc 7.2 parks the
per-lane record pointer in a32 / a33 at the top of the join block that follows the joints' InitVelocityConstraints -- a region only
the lanes that own joints (lane < n_walkers) run -- ahead of the exec restore, and step_post's body loop, which runs on all sixteen
lanes of an env, loads through the stale copies of lanes 10..15 (profiles/r06_multiwalker/c10_fused_masked_spill.txt).

A join block is recognised by its label being the target of the `s_cbranch_execz` that follows the region's `s_and_saveexec_b64 sK`
and by `s_or_b64 exec, exec, sK` being its first instruction that touches exec.  (Copies inside a THEN block, under the mask they are
meant for, are ordinary phi copies and are not reported.)"""
import re
import sys

DEF = re.compile(r"(\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))[ ,]")


def scan(path, want=""):
    text = open(path).read()
    out = []
    for fn in re.split(r"\n\t\.globl\t", text)[1:]:
        name = fn.split("\n", 1)[0].split()[0]
        if want not in name:
            continue
        lines = [l.strip() for l in fn.split("\n")]
        # regions: s_and_saveexec_b64 sK, ... ; (a few instructions) ; s_cbranch_execz LABEL
        opened = {}   # join label -> (line of the saveexec, sK)
        for i, l in enumerate(lines):
            m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]),", l)
            if not m:
                continue
            for k in range(i + 1, min(i + 6, len(lines))):
                b = re.match(r"s_cbranch_execz (\.LBB\w+)", lines[k])
                if b:
                    opened.setdefault(b.group(1), []).append((i, m.group(1)))
                    break
                if lines[k] and not lines[k].startswith(";") and not lines[k].startswith("s_") and not lines[k].startswith("v_writelane"):
                    break
        for i, l in enumerate(lines):
            m = re.match(r"(\.LBB\w+):", l)
            if not m or m.group(1) not in opened:
                continue
            copies, k = [], i + 1
            while k < len(lines):
                t = lines[k]
                if t.startswith(".LBB") or t.startswith("; %bb."):
                    break
                e = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\])", t)
                if e:
                    for start, sk in opened[m.group(1)]:
                        if sk != e.group(1):
                            continue
                        for j, c in copies:
                            src = c.split(",")[-1].strip()
                            if not re.match(r"v\d+$", src):
                                continue
                            num, inside = int(src[1:]), False
                            for q in range(j - 1, start, -1):
                                d = DEF.match(lines[q] + " ")
                                if d and not d.group(1).startswith(("global_store", "ds_write", "flat_store", "scratch_store", "buffer_store", "s_")):
                                    lo, hi = (int(d.group(5)),) * 2 if d.group(5) is not None else (int(d.group(3)), int(d.group(4)))
                                    if lo <= num <= hi:
                                        inside = True
                                        break
                            if not inside:
                                out.append((name, m.group(1), c, lines[start], k - i))
                    break
                if re.search(r"\bexec\b", t) and not t.startswith(";"):
                    break
                if t.startswith("v_accvgpr_write_b32"):
                    copies.append((k, t))
                k += 1
    return out


if __name__ == "__main__":
    hits = scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    by_kernel = {}
    for name, block, c, opener, dist in hits:
        by_kernel.setdefault(name, []).append((block, c, opener))
    for name, v in by_kernel.items():
        print("%s: %d live-through value(s) copied under the narrowed exec mask of a join" % (name, len(v)))
        for block, c, opener in v[:16]:
            print("    join %-12s %-34s   region opened by: %s" % (block, c, opener))
    if not hits:
        print("clean: no spill copy of a live-through value ahead of a join's exec restore")
    sys.exit(1 if hits else 0)
