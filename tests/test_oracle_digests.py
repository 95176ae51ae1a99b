"""The oracle's outputs on the rows without an external byte truth must not drift unnoticed (tests/golden/make_oracle_digests.py)."""
import importlib.util
import json
import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_outputs_match_the_committed_digests():
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(GOLD, "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(GOLD, "oracle_digests.json")))["digests"]
    got = mod.digests()
    assert sorted(got) == sorted(want)
    moved = {k: (want[k], got[k]) for k in want if want[k] != got[k]}
    assert not moved, f"the oracle changed its output for {sorted(moved)}: if intended, regenerate tests/golden/oracle_digests.json and say why in the commit"
