"""CPU test (-m "not gpu"): invariants of the BUILT gfx950 code objects that the hand-written parts of the kernels rely on.

* The step kernel of the Pursuit fast path issues its record prefetch through inline asm and waits for it with an exact
  s_waitcnt (pursuit_wave.hpp, "exact-wait prefetch").  The compiler does not know those registers are in flight, so it must never
  spill or reload them: the kernels must have NO scratch and no VGPR spills.
* pursuit_wave / pursuit_group / waterworld / hostage kernels read launch parameters from the kernel-argument segment through a
  struct {Dev d; IO io;} view (cold_args(), ww_args(), hw_args()): the second by-value argument must start where that view says.
* The specialised Waterworld / hostage instantiations were tuned to an occupancy at which they do not spill (spill stores reach HBM).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(HERE), "madrl_amd", "libmadrl_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels():
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(SO) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("built library or llvm tools not available")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(SO, tmp)
        subprocess.run([objdump, "--offloading", "libmadrl_hip.so"], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if not f.endswith("gfx950"):
                continue
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                args = [(int(o), int(s)) for o, s in re.findall(r"- \.offset:\s+(\d+)\s+\.size:\s+(\d+)\s+\.value_kind:\s+by_value", blk)]
                out[name.group(1)] = dict(scratch=int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)),
                                          vgpr_spills=int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)),
                                          vgprs=int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)), args=args)
    assert out, "no gfx950 kernels found in the library"
    return out


def test_prefetch_kernels_never_spill():
    ks = {n: k for n, k in _kernels().items() if "pursuit_wave_kernel" in n and "ELi1ELb0ELb0EEE" in n}   # MODE 1, INJECT false, CTRL false
    assert len(ks) >= 11
    for n, k in ks.items():
        assert k["scratch"] == 0 and k["vgpr_spills"] == 0, (n, k)


def test_kernarg_views_match_the_argument_layout():
    ks = _kernels()
    checked = 0
    for n, k in ks.items():
        if any(t in n for t in ("pursuit_wave_kernel", "pursuit_group_kernel", "waterworld_kernel", "hostage_kernel")):
            (o0, s0), (o1, _s1) = k["args"][:2]
            assert o0 == 0 and o1 == (s0 + 7) // 8 * 8, (n, k["args"])   # struct {Dev d; IO io;}: io follows d, 8-byte aligned
            checked += 1
    assert checked >= 40


def test_specialised_particle_kernels_do_not_spill():
    ks = _kernels()
    spec = {n: k for n, k in ks.items() if ("waterworld_kernelILi" in n and "ELi0ELi0ELi0ELi0E" not in n) or
            ("hostage_kernelILi" in n and "ELi0ELi0ELi0ELi0E" not in n)}
    assert len(spec) >= 10
    for n, k in spec.items():
        assert k["scratch"] == 0 and k["vgpr_spills"] == 0, (n, k)


def test_multiwalker_launches_have_no_scratch():
    """No MultiWalker kernel touches scratch memory.  The 180 + 60 sweep loops of the island solver keep every joint constant and three
    manifolds per lane in registers (512 unified VGPRs at one wavefront per SIMD); the narrow phase, the GJK proxies and the observation use
    named members and value-wise selects instead of indexed local arrays (multiwalker_toi.hpp V2x5).  Besides the latency of scratch
    accesses inside GJK, a kernel with a few hundred bytes of scratch per lane made the runtime re-allocate scratch around every other
    kernel on the stream: +3 ms per step next to a policy's torch kernels."""
    ks = {n: k for n, k in _kernels().items() if "mw_step_kernel" in n}
    # three capacity classes (mwk_c4 / _c8 / _c10: 4 / 8 / 16 lanes per env) x (collide | solve | continuous pass | the whole step in one
    # launch -- since round 6 for the sixteen-lane class too, see test_no_spill_copy_runs_under_a_narrowed_exec_mask)
    assert len(ks) == 12 and sum("mwk_c10" in n for n in ks) == 4, sorted(ks)
    for n, k in ks.items():
        # (.vgpr_spill_count may be non-zero with no private segment: at one wavefront per SIMD the allocator parks values in the 256
        # accumulation registers, a register copy each way -- what must not happen is a private segment, i.e. memory)
        assert k["scratch"] == 0, (n, k)


def test_no_spill_copy_runs_under_a_narrowed_exec_mask():
    """hipcc 7.2 may split the live range of a per-lane value at a control-flow join and emit the copy (VGPR -> accumulation register) at
    the head of the join block, AHEAD of the `s_or_b64 exec` that re-enables the lanes the region had masked off; those lanes never get
    their copy and read a stale register afterwards.  That is what made the one-launch MultiWalker kernel of the sixteen-lane class fault
    (profiles/r05_multiwalker/rocgdb_c10_fused.txt, profiles/r06_multiwalker/c10_fused_masked_spill.txt).  Round 6 removed the values
    that were live across that join (the record is found again from the lane id after the sweeps: GroupPar::rec_again) -- this test keeps
    it that way: the three capacity classes are compiled to assembly with the build's own flags and scanned (scripts/find_masked_spills.py)."""
    import importlib.util
    import sys
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(HERE)
    sys.path.insert(0, root)
    from madrl_amd import build as B
    if not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("find_masked_spills", os.path.join(root, "scripts", "find_masked_spills.py"))
    fms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fms)
    with tempfile.TemporaryDirectory() as tmp:
        def asm(c):
            out = os.path.join(tmp, c + ".s")
            subprocess.run([B.HIPCC] + [f for f in B.FLAGS if f != "-Wall"] + B.EXTRA_FLAGS["multiwalker_c"] +
                           ["--cuda-device-only", "-S", os.path.join(B.CSRC, "multiwalker_%s.hip" % c), "-o", out], check=True, capture_output=True)
            return out
        with ThreadPoolExecutor(3) as ex:
            files = list(ex.map(asm, ("c4", "c8", "c10")))
        for f in files:
            text = open(f).read()
            assert text.count("mw_step_kernelILi7E") > 0, "the one-launch kernel is missing from " + f
            hits = fms.scan(f)
            assert not hits, [(h[0], h[1], h[2]) for h in hits]
