"""INTEGRATION.md section 2 is executable: the binding a maintainer copies from the document must be accepted by the library
(round 3: the documented struct had fallen behind the header and madrl_pursuit_create answered MADRL_EINVAL)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The binding"):text.index("## 3. Reference-side change")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    return code.replace('C.CDLL("madrl_amd/libmadrl_hip.so")', 'C.CDLL(%r)' % os.path.join(ROOT, "madrl_amd", "libmadrl_hip.so"))


def test_documented_struct_is_the_library_struct():
    """CPU: the class definition and the cfg = ... line of the document, checked field by field against madrl_amd/_lib.py
    (which tests/test_abi.py checks against the header) and accepted by the library's validation"""
    from madrl_amd import _lib
    code = _snippet()
    head = code[code.index("class PursuitConfig"):code.index("N = 65536")]
    ns = {"C": C}
    exec(head, ns)
    Doc = ns["PursuitConfig"]
    assert [(n, t) for n, t in Doc._fields_] == [(n, t) for n, t in _lib.PursuitConfig._fields_]
    assert C.sizeof(Doc) == C.sizeof(_lib.PursuitConfig)
    L = _lib.lib()
    b = C.c_uint64()
    assert L.madrl_pursuit_state_bytes(C.byref(ns["cfg"]), 65536, C.byref(b)) == 0, L.madrl_last_error()
    assert b.value == 65536 * (112 + 256 + 4)   # records, stale-zero masks, flag plane


@pytest.mark.gpu
def test_documented_binding_runs_as_written():
    import torch
    from madrl_amd.maps import rectangle_map
    code = _snippet()
    N = 512
    code = code.replace("N = 65536", "N = %d" % N)
    ns = {"map_pool_int8": np.ascontiguousarray(np.asarray(rectangle_map(16, 16), np.int8)[None]),
          "actions_i32": torch.randint(0, 5, (N, 8), dtype=torch.int32, device="cuda")}
    exec(code, ns)
    torch.cuda.synchronize()
    assert ns["rc"] == 0
    obs = ns["obs"].cpu().numpy()
    assert np.isfinite(obs).all() and (obs[:, :, -1] == (np.arange(8) / 8.0).astype(np.float32)).all()   # the id feature i / n_agents
    assert ns["done"].cpu().numpy().max() <= 3
    ns["L"].madrl_pursuit_destroy(ns["h"])
