"""Build libmadrl_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python -m madrl_amd.build [--force]
    python -m madrl_amd.build --pursuit-shape XS YS N_PURSUERS N_EVADERS OBS_RANGE FLATTEN      # give this shape the fast path, rebuild
    python -m madrl_amd.build --waterworld-shape N_PURSUERS N_EVADERS N_POISON N_SENSORS [OBS_DIM]

The fast paths (one wavefront -- or a group of wavefronts -- per env, everything about the shape a compile-time constant) exist for the
shapes listed in csrc/*_specializations.def: the BASELINE configurations, the reference's own runner / script defaults, the test shapes.
Any other shape runs on the generic kernels, at about half the speed.  The two options above append a line to
csrc/*_specializations.local.def (git-ignored, included after the committed list) and rebuild the one object that changed (~30 s).

-ffp-contract=off: reward arithmetic and the reset window are float64 expressions that must
round step by step like NumPy does in the reference; an FMA contraction would change bits.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libmadrl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]   # (`#pragma unroll` on loops whose bounds are only known in the specialised instantiations: the generic ones stay rolled)
# per-source additions.  multiwalker: the SLP vectorizer pairs the solver's scalar float math into packed-fp32 instructions, which
# need every loop constant replicated into register pairs -- the 180-sweep loop then runs out of VGPRs (AGPR copies, scratch)
EXTRA_FLAGS = {"multiwalker_c": ["-fno-slp-vectorize"]}   # file-name prefix -> flags (the capacity classes multiwalker_c4 / _c8 / _c10.hip)


def pursuit_fast_path(xs, ys, n_pursuers, n_evaders, obs_range, flatten, include_id=True):
    """-> ("X", None) one wavefront per env, ("XG", NW) NW wavefronts per env, or (None, why not): the static_asserts of
    pursuit_wave.hpp / pursuit_group.hpp, evaluated here so that a shape that cannot have a fast path is refused before it breaks the build"""
    P, E, R = int(n_pursuers), int(n_evaders), int(obs_range)
    A = P + E
    if R % 2 == 0:
        return None, "even obs_range"
    if flatten and not include_id:
        return None, "flatten without the id: rows are not whole float4s"
    if P > 64 or E > 64 or A > 128:
        return None, "more than 64 pursuers or evaders (or 128 agents)"
    D = 3 * R * R + 1 if flatten else 4 * R * R
    if D % 4:
        return None, "observation row is not a whole number of float4"
    # one wavefront per env when the agents fit its lanes AND the row fits 8 float4 slots per lane (slot constants in registers,
    # pursuit_wave.hpp); otherwise two wavefronts (pursuit_group.hpp) -- four for long rows, whose slot constants come from an LDS table ("LONG ROWS": more than 8 slots per thread of two wavefronts), up to 32 slots per thread
    slots = lambda n: (P * (D // 4) + 64 * n - 1) // (64 * n)
    nw = 1 if (A <= 64 and slots(1) <= 8) else (2 if slots(2) <= 8 else 4)   # long rows: four wavefronts share the env's LDS (occupancy is LDS-bound there)
    if slots(nw) > 32:
        return None, "more than 32 float4 slots per thread (n_pursuers x row length too large)"
    pad = max((R - 1) // 2, 1)
    gsz = ((xs + 2 * pad) * (ys + 2 * pad) + 3) // 4 * 4
    tabled = slots(nw) > 8
    if (3 * gsz + 2 + P + 72 + (xs * ys + 3) // 4 + 2 * P + 16 + (2 * (D // 4) if tabled else 0)) * 4 > 64 * 1024:
        return None, "map too large for the LDS layers"
    if tabled and 3 * gsz + 2 + P >= 32768:
        return None, "map too large for the 15-bit offsets of the long-row table"
    ngw, ntw = max((E + 31) // 32, 1), (A + 31) // 32
    rec = ((16 + 2 * A + 3) // 4 * 4 + 4 * ngw + 4 * ntw + 15) // 16 * 16
    if rec > 256:
        return None, "state record above 256 bytes"
    return ("X", None) if nw == 1 else ("XG", nw)


def _append_local(def_name, line):
    path = os.path.join(CSRC, def_name.replace(".def", ".local.def"))
    committed = open(os.path.join(CSRC, def_name)).read() + (open(path).read() if os.path.exists(path) else "")
    norm = lambda t: "".join(t.split())
    if norm(line.split("//")[0]) in norm(committed):
        return False
    with open(path, "a") as f:
        f.write(line + "\n")
    return True


def add_pursuit_shape(xs, ys, n_pursuers, n_evaders, obs_range, flatten):
    kind, nw = pursuit_fast_path(xs, ys, n_pursuers, n_evaders, obs_range, flatten)
    if kind is None:
        raise ValueError("no fast path for this PursuitEvade shape: %s (it runs on the generic kernel)" % nw)
    args = "%d, %d, %d, %d, %d, %d" % (xs, ys, n_pursuers, n_evaders, obs_range, int(bool(flatten)))
    return _append_local("pursuit_specializations.def", "X(%s)   // added by madrl_amd.build" % args if kind == "X" else
                         "XG(%s, %d)   // added by madrl_amd.build" % (args, nw))


def add_waterworld_shape(n_pursuers, n_evaders, n_poison, n_sensors, obs_dim=None):
    if obs_dim is None:
        obs_dim = n_sensors * 7 + 2 + 1          # speed features and the agent id (the reference's defaults)
    if n_pursuers + n_evaders + n_poison > 62 or 2 * n_pursuers > 64 or not 1 <= n_sensors <= 256:
        raise ValueError("no Waterworld kernel for this shape at all (madrl_waterworld_create refuses it)")
    return _append_local("waterworld_specializations.def", "X(%d, %d, %d, %d, %d)   // added by madrl_amd.build" % (n_pursuers, n_evaders, n_poison, n_sensors, obs_dim))


def specialised_shapes(def_name):
    """the X(...) / XG(...) argument tuples of a csrc/*_specializations.def list and of its git-ignored .local.def companion"""
    out = set()
    for path in (os.path.join(CSRC, def_name), os.path.join(CSRC, def_name.replace(".def", ".local.def"))):
        if os.path.exists(path):
            for m in re.finditer(r"^\s*XG?\(([^)]*)\)", open(path).read(), re.M):
                out.add(tuple(int(v) for v in m.group(1).split(",")))
    return out


def waterworld_is_specialised(n_pursuers, n_evaders, n_poison, n_sensors, obs_dim):
    return (int(n_pursuers), int(n_evaders), int(n_poison), int(n_sensors), int(obs_dim)) in specialised_shapes("waterworld_specializations.def")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(src, seen=None):
    """the files one source depends on: itself and, recursively, every `#include "..."` in it that exists (the git-ignored *.local.def
    lists are behind __has_include: they count once they exist) -- so appending a Pursuit shape re-compiles pursuit.hip and nothing else"""
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    for inc in _INC.findall(open(src).read()):
        _deps(os.path.normpath(os.path.join(os.path.dirname(src), inc)), seen)
    return seen


def _compile(src, obj, verbose):
    cmd = [HIPCC] + FLAGS + [f for pat, fl in EXTRA_FLAGS.items() if os.path.basename(src).startswith(pat) for f in fl] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    objs, stale = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src)):
            stale.append((src, obj))
    if stale:   # the objects are independent: compile them side by side (the three MultiWalker classes take a minute each)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 1)) as ex:
            for f in [ex.submit(_compile, src, obj, verbose) for src, obj in stale]:
                f.result()
    for f in os.listdir(CSRC):   # objects whose source is gone (a renamed file) must not be linked
        if f.endswith(".o") and os.path.join(CSRC, f) not in objs:
            os.remove(os.path.join(CSRC, f))
    if force or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    argv = sys.argv[1:]
    for flag, fn, lo, hi in (("--pursuit-shape", add_pursuit_shape, 6, 6), ("--waterworld-shape", add_waterworld_shape, 4, 5)):
        while flag in argv:
            i = argv.index(flag)
            vals = []
            while i + 1 + len(vals) < len(argv) and len(vals) < hi and argv[i + 1 + len(vals)].lstrip("-").isdigit():
                vals.append(int(argv[i + 1 + len(vals)]))
            if len(vals) < lo:
                raise SystemExit("%s takes %d%s integers" % (flag, lo, "" if lo == hi else " or %d" % hi))
            print("%s %s: %s" % (flag, vals, "added" if fn(*vals) else "already specialised"))
            del argv[i:i + 1 + len(vals)]
    print(build(force="--force" in argv, verbose=True))
