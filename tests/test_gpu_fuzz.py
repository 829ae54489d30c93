"""A short run of the randomised GPU-vs-oracle sweep (tests/fuzz_parity.py; the long runs are recorded under profiles/)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_parity_short():
    pytest.importorskip("torch")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "12", "2026"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "cases bit-identical" in p.stdout
