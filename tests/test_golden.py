"""The CPU oracle against the committed golden digests (tests/golden/oracle_digests.json, made by
tests/golden/make_golden.py)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_matches_golden_digests():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")))
    got = mg.compute()
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], f"oracle output changed for {k}"
