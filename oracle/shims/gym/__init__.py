"""Minimal stand-in for the `gym` package (absent in this image) so that the
UNMODIFIED reference under /root/reference imports.  Test infrastructure only:
used by oracle/ref_loader.py to generate golden vectors.  Not shipped, not
imported by madrl_amd."""
from . import spaces, error  # noqa: F401
