"""The oracle against its own committed output digests (tests/golden/oracle_digests.json).  Not a
pin to the reference (see the generator's header): a tripwire for unintended arithmetic changes."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_outputs_match_committed_digests(oracle):
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(HERE, "golden", "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))
    got = mod.digests()
    assert sorted(got) == sorted(want)
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, f"oracle output changed for: {bad}"
