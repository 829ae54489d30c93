"""The oracle must not drift: its outputs on seeded inputs are pinned by hash (tests/golden/oracle_hashes.json, regenerated
only deliberately with tests/golden/make_oracle_hashes.py). CPU only."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_outputs_match_committed_hashes(oracle):
    spec = importlib.util.spec_from_file_location("make_oracle_hashes", os.path.join(HERE, "golden", "make_oracle_hashes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "oracle_hashes.json")))
    got = {name: mod.h(arr) for name, arr in mod.cases()}
    assert set(got) == set(want)
    bad = [n for n in got if got[n] != want[n]]
    assert not bad, f"oracle output changed for {bad}: if intended, rerun tests/golden/make_oracle_hashes.py"
