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
    """`python bench.py --gpus 2` with NO launcher around it — the shape of the driver's own command — starts its two ranks itself
    (rendezvous on 127.0.0.1, barriers, max-over-ranks clock, one JSON line from rank 0). Both ranks share cuda:0 here and gloo carries
    the control-plane collectives (ZG_BENCH_SHARED_GPU=1, a test hook the driver never sets)."""
    import json
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ZG_BENCH_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "100"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 100 and res["scaling"] == "weak" and res["value"] > 0
    assert res["ranks"]["ranks_seen"] == 2 and len(res["ranks"]["ms_per_step_per_rank"]) == 2 and len(res["ranks"]["devices"]) == 2
    assert abs(max(res["ranks"]["ms_per_step_per_rank"]) - res["ms_per_step"]) < 1e-3  # the slowest rank is the clock
    # rank 0 reports its own GPU's roofline at every N (so that N = 1 here and the driver's first scaling point agree by construction) and the
    # N-GPU run carries BASELINE configs[4]'s scatter / compute / gather leg without a flag; extras and the CPU baseline stay N = 1 legs
    assert res["roofline"]["bound"] == "hbm" and res["roofline"]["achieved"] > 0 and "extras" not in res and "cpu_baseline" not in res
    sg = res["scatter_gather_config5"]
    assert sg["frames"] == 256 and sg["kernel_only"]["Mpixels/s"] > 0 and sg["host_staged_pcie"]["Mpixels/s"] > 0
    assert "skipped" in sg["end_to_end_xgmi"]  # every rank shares cuda:0 under the test hook: RCCL needs one GPU per rank


def test_bench_counts_its_rccl_ranks_and_times_the_three_distribution_routes():
    """One GPU, `--scatter-gather`: the process group is RCCL (world 1) and the JSON says so; BASELINE configs[4] comes out three ways
    (kernel only, end to end through the communicator — the shard looped back —, host-staged over PCIe)."""
    import json
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "100", "--scatter-gather", "--no-extras", "--no-cpu-baseline", "--no-live-traffic"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert res["ranks"]["backend"].startswith("nccl") and res["ranks"]["rccl_ranks"] == 1
    sg = res["scatter_gather_config5"]
    assert "error" not in sg, sg
    for leg in ("kernel_only", "end_to_end_xgmi", "end_to_end_xgmi_unchunked", "host_staged_pcie"):
        assert sg[leg]["Mpixels/s"] > 0, (leg, sg)
    assert sg["kernel_only"]["Mpixels/s"] > sg["end_to_end_xgmi"]["Mpixels/s"]
