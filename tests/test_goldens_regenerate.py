"""CPU test, build container only: every file under tests/golden/ is what its generator writes TODAY when it runs the unmodified reference
from /root/reference -- byte for byte.  (The goldens are the anchor of every parity claim; this is the check that they are the reference's
output and not something edited afterwards.  /root/reference does not travel to the GPU box: the test skips there.)  The one family left
out is multiwalker_box2d_*: it needs pybox2d, which this image does not have (PARITY UNPINNED, tests/test_oracle_multiwalker_golden.py)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
GENERATORS = [("make_golden_reset_hist.py", []), ("make_golden_multiwalker.py", ["--envlayer"]), ("make_golden_pursuit.py", []), ("make_golden_pursuit_fuzz.py", []),
              ("make_golden_waterworld.py", []), ("make_golden_waterworld_fuzz.py", []), ("make_golden_hostage.py", []), ("make_golden_hostage_fuzz.py", []),
              ("make_golden_pursuit_evader.py", []), ("make_golden_callers.py", []), ("make_golden_wrappers.py", []), ("make_golden_heuristics.py", []),
              ("make_golden_curriculum.py", [])]


@pytest.mark.skipif(not os.path.isdir("/root/reference/madrl_environments"), reason="the reference tree is not on this machine")
def test_every_golden_file_regenerates_byte_for_byte(tmp_path):
    env = dict(os.environ, MADRL_GOLDEN_OUT=str(tmp_path))

    def run(g):
        return g, subprocess.run([sys.executable, os.path.join(ROOT, "oracle", g[0])] + g[1], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)

    with ThreadPoolExecutor(4) as ex:
        for g, r in ex.map(run, GENERATORS):
            assert r.returncode == 0, "%s failed:\n%s" % (g[0], r.stderr[-2000:])
    made = sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*")))
    have = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*")))
    assert made == have, "generated but not committed: %s; committed but not generated: %s" % (sorted(set(made) - set(have)), sorted(set(have) - set(made)))
    differ = [n for n in made if open(os.path.join(str(tmp_path), n), "rb").read() != open(os.path.join(GOLDEN, n), "rb").read()]
    assert not differ, "regenerated files differ from the committed ones: %s" % differ
