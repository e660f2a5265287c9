"""CPU tests (-m "not gpu"): the C-ABI shared library loads without a GPU, exports every
symbol include/madrl_hip.h declares, and its host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "madrl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(madrl_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from madrl_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), "libmadrl_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "madrl_amd/_lib.py has no signature for %s" % n
    assert L.madrl_abi_version() == _lib.ABI_VERSION == 7
    hdr = open(os.path.join(ROOT, "include", "madrl_hip.h")).read()
    assert int(re.search(r"#define MADRL_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    words = re.search(r"#define MADRL_POLICY_COUNTER_WORDS \((\d+) \* (\d+)\)", hdr)
    assert int(words.group(1)) * int(words.group(2)) == _lib.POLICY_COUNTER_WORDS


def test_host_philox_matches_published_vectors():
    from madrl_amd import _lib
    L = _lib.lib()
    out = np.zeros(4, np.uint32)
    ctr = np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32)
    key = np.array([0xa4093822, 0x299f31d0], np.uint32)
    L.madrl_philox4x32_10(ctr.ctypes.data_as(C.c_void_p), key.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert [hex(v) for v in out] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def _cfg(**over):
    from madrl_amd import _lib
    c = _lib.PursuitConfig()
    c.struct_size = C.sizeof(_lib.PursuitConfig)
    c.xs = c.ys = 16
    c.n_pursuers, c.n_evaders, c.obs_range, c.n_catch = 8, 30, 7, 2
    c.surround = c.flatten = c.include_id = 1
    c.n_maps = 1
    c.layer_norm, c.constraint_window = 10.0, 1.0
    for k, v in over.items():
        setattr(c, k, v)
    return c


def test_obs_dim_and_state_bytes():
    from madrl_amd import _lib
    L = _lib.lib()
    d = C.c_int32()
    assert L.madrl_pursuit_obs_dim(C.byref(_cfg()), C.byref(d)) == 0 and d.value == 148
    assert L.madrl_pursuit_obs_dim(C.byref(_cfg(flatten=0)), C.byref(d)) == 0 and d.value == 196
    assert L.madrl_pursuit_obs_dim(C.byref(_cfg(include_id=0)), C.byref(d)) == 0 and d.value == 147
    b = C.c_uint64()
    assert L.madrl_pursuit_state_bytes(C.byref(_cfg()), 65536, C.byref(b)) == 0
    r = C.c_int32()
    assert L.madrl_pursuit_record_bytes(C.byref(_cfg()), C.byref(r)) == 0
    assert r.value == 112  # 16 B header + 76 B positions + masks, 16-B aligned
    # records, then the fast path's stale-zero masks (256 B per env), then the flag plane (ABI 7: one dword per env): all caller-owned
    assert b.value == 65536 * (112 + 256 + 4)
    f = C.c_uint64()
    assert L.madrl_pursuit_flags_offset(C.byref(_cfg()), 65536, C.byref(f)) == 0 and f.value == 65536 * (112 + 256)
    assert L.madrl_pursuit_state_bytes(C.byref(_cfg(n_pursuers=7)), 65536, C.byref(b)) == 0 and b.value == 65536 * (112 + 4)  # no fast path
    assert L.madrl_pursuit_flags_offset(C.byref(_cfg(n_pursuers=7)), 65536, C.byref(f)) == 0 and f.value == 65536 * 112
    assert L.madrl_pursuit_state_bytes(C.byref(_cfg(control_evaders=1)), 1000, C.byref(b)) == 0 and b.value == 112128 + 1000 * 256 + 4000  # records padded to 256 B, then the masks, then the flags
    assert L.madrl_pursuit_state_bytes(C.byref(_cfg(n_pursuers=16, n_evaders=60, xs=32, ys=32, control_evaders=1)), 1000, C.byref(b)) == 0 and (b.value - 4000) % 256 == 0 \
        and b.value < 1000 * 512   # evader control above one wavefront of agents: generic kernel only, no masks


def test_invalid_configs_are_rejected_with_a_message():
    from madrl_amd import _lib
    L = _lib.lib()
    d = C.c_int32()
    for bad in (dict(struct_size=4), dict(xs=0), dict(n_pursuers=0), dict(n_evaders=1024), dict(layer_norm=0.0),
                dict(constraint_window=0.0), dict(obs_range=0)):
        rc = L.madrl_pursuit_obs_dim(C.byref(_cfg(**bad)), C.byref(d))
        assert rc == -1, bad
        assert len(L.madrl_last_error()) > 0


def test_product_never_touches_the_oracle():
    """madrl_amd must not import / include / link / call anything under oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "madrl_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#\s*include\s*[<\"][^>\"]*oracle)|(oracle/)|(madrl_oracle)|(\bpo_[a-z_]+\s*\()", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".def")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), (dirpath, f, bad.search(src).group(0))
    # and the shared library does not link it
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libmadrl_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from madrl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MadrlError):
        _lib.lib()


def test_env_refuses_cpu_device():
    from madrl_amd import _lib
    from madrl_amd.pursuit import BatchedPursuitEvade
    from madrl_amd.maps import rectangle_map
    with pytest.raises(_lib.MadrlError):
        BatchedPursuitEvade([rectangle_map(16, 16)], n_envs=4, device="cpu", n_pursuers=2, n_evaders=2)


def test_rectangle_map_matches_reference_fixture():
    """maps.rectangle_map restates utils/TwoDMaps.py:8-22; golden files carry the reference's."""
    from madrl_amd.maps import rectangle_map
    g = np.load(os.path.join(ROOT, "tests", "golden", "pursuit_c1_surround_local.npz"))
    np.testing.assert_array_equal(rectangle_map(16, 16), g["maps"][0])
    g = np.load(os.path.join(ROOT, "tests", "golden", "pursuit_c5_32x32.npz"))
    np.testing.assert_array_equal(rectangle_map(32, 32), g["maps"][0])
    g = np.load(os.path.join(ROOT, "tests", "golden", "pursuit_nonsquare_12x20.npz"))
    np.testing.assert_array_equal(rectangle_map(12, 20), g["maps"][0])


def test_batch_size_limits_are_checked_at_create():
    """the kernels index envs with 32-bit integers: every create() refuses a batch that would overflow them (no GPU needed:
    the check precedes any HIP call)"""
    from madrl_amd import _lib
    L = _lib.lib()
    sens = np.zeros((30, 2))
    dummy = C.c_void_p(16)
    out = C.c_void_p()
    wc = _lib.WaterworldConfig()
    wc.struct_size = C.sizeof(_lib.WaterworldConfig)
    wc.n_pursuers, wc.n_evaders, wc.n_coop, wc.n_poison, wc.n_sensors, wc.addid, wc.speed_features, wc.obstacle_fixed = 5, 10, 2, 10, 30, 1, 1, 1
    wc.radius, wc.obstacle_radius, wc.ev_speed, wc.poison_speed, wc.sensor_range, wc.action_scale = 0.015, 0.2, 0.01, 0.01, 0.2, 0.01
    assert L.madrl_waterworld_create(C.byref(wc), sens.ctypes.data_as(C.c_void_p), 2**31 - 10, 0, dummy, C.byref(out)) == -1
    assert b"too large" in L.madrl_last_error()
    hc = _lib.HostageConfig()
    hc.struct_size = C.sizeof(_lib.HostageConfig)
    hc.n_good, hc.n_hostages, hc.n_bad, hc.n_coop_save, hc.n_coop_avoid, hc.n_sensors, hc.addid, hc.key_fixed = 3, 10, 5, 2, 2, 30, 1, 1
    hc.radius, hc.bad_speed, hc.sensor_range, hc.action_scale, hc.bomb_radius, hc.key_radius = 0.015, 0.01, 0.2, 0.01, 0.03, 0.0225
    assert L.madrl_hostage_create(C.byref(hc), sens.ctypes.data_as(C.c_void_p), 2**31 - 10, 0, dummy, C.byref(out)) == -1
    assert b"too large" in L.madrl_last_error()


def test_which_shapes_can_have_a_fast_path_and_how_they_are_added(tmp_path, monkeypatch):
    """madrl_amd.build.pursuit_fast_path mirrors the static_asserts of pursuit_wave.hpp / pursuit_group.hpp (a shape they refuse must be refused
    before it breaks the build); --pursuit-shape / --waterworld-shape append to a git-ignored *.local.def that the sources include"""
    from madrl_amd import build as b
    assert b.pursuit_fast_path(16, 16, 8, 30, 7, 1) == ("X", None) and b.pursuit_fast_path(32, 32, 16, 60, 7, 1) == ("XG", 2)
    assert b.pursuit_fast_path(16, 16, 8, 30, 7, 0) == ("X", None) and b.pursuit_fast_path(20, 8, 7, 1, 3, 0) == ("X", None)
    # the authors' own training shapes (runners/old/rllab/pursuit.sh:1, runners/old/rltools/pursuit.sh:1): 22 float4 slots per thread on two
    # wavefronts -- also at 30 v 30, where the agents alone would fit one
    assert b.pursuit_fast_path(32, 32, 30, 50, 11, 1) == ("XG", 4) and b.pursuit_fast_path(32, 32, 30, 30, 11, 1) == ("XG", 4)   # long rows: four wavefronts
    assert b.pursuit_fast_path(5, 10, 16, 7, 7, 1) == ("XG", 2)   # 23 agents, but 10 slots per lane of one wavefront: two wavefronts, 5 each
    for shape, why in (((10, 10, 4, 4, 4, 1), "even"), ((128, 128, 100, 300, 21, 0), "more than 64"), ((16, 16, 64, 10, 21, 0), "slots"),
                       ((200, 200, 8, 30, 7, 1), "LDS"), ((24, 24, 70, 58, 3, 1), "more than 64")):
        kind, reason = b.pursuit_fast_path(*shape)
        assert kind is None and why in reason, (shape, reason)
    assert b.pursuit_fast_path(16, 16, 8, 30, 7, 1, include_id=False)[0] is None
    # every committed line passes its own check
    for line in open(os.path.join(ROOT, "madrl_amd", "csrc", "pursuit_specializations.def")):
        m = re.match(r"\s*(XG?)\(([^)]*)\)", line)
        if m:
            v = [int(x) for x in m.group(2).split(",")]
            kind, nw = b.pursuit_fast_path(*v[:6])
            assert kind == m.group(1) and (kind == "X" or nw == v[6]), line
    # appending: a new shape lands in the local file once, a committed one not at all
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for f in ("pursuit_specializations.def", "waterworld_specializations.def"):
        (csrc / f).write_text(open(os.path.join(ROOT, "madrl_amd", "csrc", f)).read())
    monkeypatch.setattr(b, "CSRC", str(csrc))
    assert b.add_pursuit_shape(20, 20, 6, 10, 5, 1) is True and b.add_pursuit_shape(20, 20, 6, 10, 5, 1) is False
    assert b.add_pursuit_shape(16, 16, 8, 30, 7, 1) is False and b.add_waterworld_shape(4, 8, 6, 24) is True and b.add_waterworld_shape(5, 10, 10, 30) is False
    assert (csrc / "pursuit_specializations.local.def").read_text().startswith("X(20, 20, 6, 10, 5, 1)")
    assert (csrc / "waterworld_specializations.local.def").read_text().startswith("X(4, 8, 6, 24, 171)")
    # ... and is then a specialised shape for the hint a large generic-kernel batch gives (madrl_amd/waterworld.py _hint_fast_path)
    assert b.waterworld_is_specialised(4, 8, 6, 24, 171) and b.waterworld_is_specialised(5, 10, 10, 30, 213) and not b.waterworld_is_specialised(4, 8, 6, 25, 178)
    with pytest.raises(ValueError):
        b.add_pursuit_shape(10, 10, 4, 4, 4, 1)
    for src in ("pursuit.hip", "waterworld.hip"):
        assert "specializations.local.def" in open(os.path.join(ROOT, "madrl_amd", "csrc", src)).read()
