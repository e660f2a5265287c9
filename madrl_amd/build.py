"""Build libmadrl_hip.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python -m madrl_amd.build [--force]

-ffp-contract=off: reward arithmetic and the reset window are float64 expressions that must
round step by step like NumPy does in the reference; an FMA contraction would change bits.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libmadrl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# per-source additions.  multiwalker: the SLP vectorizer pairs the solver's scalar float math into packed-fp32 instructions, which
# need every loop constant replicated into register pairs -- the 180-sweep loop then runs out of VGPRs (AGPR copies, scratch)
EXTRA_FLAGS = {"multiwalker.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    inc = os.path.join(os.path.dirname(HERE), "include")
    out = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".def"))]
    out += [os.path.join(inc, f) for f in os.listdir(inc)]
    return out


def build(force=False, verbose=False):
    objs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        stale = force or not os.path.exists(obj) or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps())
        if stale:
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
