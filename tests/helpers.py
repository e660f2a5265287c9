"""Shared helpers for the parity tests (golden replay, oracle <-> HIP state conversion)."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pursuit_golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "pursuit_*.npz")))


def golden_id(path):
    return os.path.basename(path)[:-4]
