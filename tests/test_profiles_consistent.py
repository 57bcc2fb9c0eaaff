"""The committed bench lines and the committed counter profiles must tell the same story: every roofline field of
profiles/r02/bench_*.json is re-derived from the per-dispatch counter values of the profile of the same command line
(tools/recompute_roofline.py) -- a stale profile, a line made before its profile, or a fraction above 1 fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("round_dir", ["r02", "r03"])
def test_roofline_fields_follow_from_the_committed_counters(round_dir):
    d = os.path.join(ROOT, "profiles", round_dir)
    import glob
    if not glob.glob(os.path.join(d, "pmc_bench_*.json")):
        pytest.skip("no counter profiles committed under profiles/%s yet" % round_dir)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "recompute_roofline.py"), d],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("frac") >= 3, r.stdout  # default line, driver line, APD line
