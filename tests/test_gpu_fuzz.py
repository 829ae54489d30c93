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


def test_bench_two_rank_path_on_one_gpu():
    """bench.py's N > 1 control flow (rendezvous, barriers, max-over-ranks clock, one JSON line from rank 0) with both ranks
    on cuda:0 and gloo for the control-plane collectives (ZG_BENCH_SHARED_GPU=1, a test hook the driver never sets)."""
    import json
    pytest.importorskip("torch")
    env = dict(os.environ, ZG_BENCH_SHARED_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "100"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 100 and res["scaling"] == "weak" and res["value"] > 0
    assert "roofline" not in res  # N = 1 legs only
