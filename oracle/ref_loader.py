"""Import the UNMODIFIED reference (sisl/MADRL at /root/reference) under the
shims in oracle/shims.  TEST INFRASTRUCTURE ONLY -- used by oracle/make_golden_*.py
in the build container to produce tests/golden/*.npz.  /root/reference does not
exist on the GPU box, so nothing at test/bench run time may import this module.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("MADRL_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "madrl_environments"))


def load():
    """Returns the reference module namespace {PursuitEvade, MAWaterWorld, TwoDMaps, ...}."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")
    for p in (REFERENCE_ROOT, _SHIMS):
        if p not in sys.path:
            sys.path.insert(0, p)
    from madrl_environments.pursuit import PursuitEvade, MAWaterWorld  # noqa
    from madrl_environments.pursuit.utils import TwoDMaps, AgentLayer, Controllers  # noqa
    import madrl_environments
    return dict(PursuitEvade=PursuitEvade, MAWaterWorld=MAWaterWorld, TwoDMaps=TwoDMaps,
                madrl_environments=madrl_environments)
