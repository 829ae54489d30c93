"""zg_multi's world > 1 branches on ONE GPU.

No test box has two GPUs, so `zg_multi.cpp`'s scatter / compute / gather over several devices had never executed anywhere. Two test-only
switches change that: ZIGNAL_HIP_MULTI_VIRTUAL lets one device id be listed N times (a context of world N whose shards, staging buffers,
streams and events are all real), and ZIGNAL_HIP_RCCL_LIBRARY binds tests/c/rccl_double.cpp instead of librccl — a stand-in that pairs the
grouped ncclSend / ncclRecv calls and turns each pair into an event-ordered device copy. Worlds 2, 3 and 8 with 1, 4 and 8 pieces per shard,
from a bare C++ process and from Python, bit-compared with the one-device call; and a send that fails half-way: the call fails, the context is
poisoned, destroying it drains the streams, a fresh context works. xGMI itself stays unmeasured: the bytes never leave the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "rccl_double.cpp")
LIB = os.path.join(ROOT, "tests", "c", "librccl_double.so")
EXPORTS = ("ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclGetErrorString",
           "rccl_double_stats", "rccl_double_reset")


def _build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", LIB, SRC], check=True, capture_output=True)
    return LIB


def test_the_stand_in_builds_and_exports_what_zg_multi_binds():
    """CPU: the stand-in compiles against the HIP headers and exports every entry point load_rccl() looks up (zg_multi.cpp:60-66)."""
    lib = _build()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for name in EXPORTS:
        assert f" T {name}" in out, name
    src = open(os.path.join(ROOT, "zignal_amd", "csrc", "zg_multi.cpp")).read()
    for name in EXPORTS[:7]:
        assert f'"{name}"' in src, f"zg_multi.cpp no longer binds {name}"


def _env():
    return dict(os.environ, ZIGNAL_HIP_MULTI_VIRTUAL="1", ZIGNAL_HIP_RCCL_LIBRARY=_build())


@pytest.mark.gpu
def test_virtual_worlds_from_a_bare_cpp_process():
    from tests.test_cpp_mirror import BIN_DEV, _build as build_cpp

    build_cpp()
    out = subprocess.run([BIN_DEV, "virtual"], capture_output=True, text=True, timeout=600, env=_env())
    log = out.stdout + out.stderr
    assert out.returncode == 0 and "device image ok" in log, log
    for world in (2, 3, 8):
        for pieces in (1, 4, 8):
            assert f"multi_virtual_world{world}_{pieces}_pieces=ok" in log, log
    assert "multi_virtual_failure_injection=ok" in log, log


PY_CODE = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import zignal_amd as zg
dbl = ctypes.CDLL(os.environ["ZIGNAL_HIP_RCCL_LIBRARY"])
stats = (ctypes.c_uint64 * 5)()
rng = np.random.default_rng(11)
frames = torch.from_numpy(rng.integers(0, 256, (11, 135, 244, 4), dtype=np.uint8)).cuda()
recipes = {
    "blur+resize": [zg.Step.gaussian_blur(0.8), zg.Step.resize(67, 122, zg.Interpolation.bilinear)],
    "resize+convert": [zg.Step.resize(50, 60, zg.Interpolation.bilinear), zg.Step.convert(zg.CS_OKLAB)],
    "box+sobel": [zg.Step.box_blur(2), zg.Step.edges_sobel()],
}
want = {name: zg.Pipeline(steps).run(frames).cpu().numpy() for name, steps in recipes.items()}
torch.cuda.synchronize()
for world in (2, 3, 8):
    for chunks in (1, 4, 8):
        os.environ["ZIGNAL_HIP_MULTI_CHUNKS"] = str(chunks)
        with zg.Multi([0] * world) as ctx:
            assert ctx.device_count() == world
            for name, steps in recipes.items():
                dbl.rccl_double_reset()
                got, times = zg.Pipeline(steps).run_multi(ctx, frames)
                got = got.cpu().numpy()
                assert got.dtype == want[name].dtype and np.array_equal(got.view(np.uint8), want[name].view(np.uint8)), (world, chunks, name)
                dbl.rccl_double_stats(stats)
                assert stats[0] == stats[1] == stats[4] and stats[0] >= 2, list(stats)
        print(f"py_virtual_world{world}_{chunks}_pieces=ok")
# a failure half-way: the third send of the batch (world 3, one piece per shard: the first result on its way back)
os.environ["ZIGNAL_HIP_MULTI_CHUNKS"] = "1"
ctx = zg.Multi([0, 0, 0])
dbl.rccl_double_reset()
os.environ["RCCL_DOUBLE_FAIL_SEND"] = "3"
try:
    zg.Pipeline(recipes["blur+resize"]).run_multi(ctx, frames)
    raise SystemExit("the injected failure did not surface")
except zg.ZignalError as e:
    assert "RCCL error" in str(e), e
os.environ.pop("RCCL_DOUBLE_FAIL_SEND")
try:
    zg.Pipeline(recipes["blur+resize"]).run_multi(ctx, frames)
    raise SystemExit("a poisoned context accepted work")
except zg.ZignalError as e:
    assert "failed half-way" in str(e), e
ctx.close()
with zg.Multi([0, 0, 0]) as ctx:
    got, _ = zg.Pipeline(recipes["blur+resize"]).run_multi(ctx, frames)
    assert np.array_equal(got.cpu().numpy(), want["blur+resize"])
# frames that do not live on the context's root device are refused before anything is enqueued (ADVICE r05)
with zg.Multi([0, 0]) as ctx:
    assert ctx.root_device == 0
    try:
        zg.Pipeline(recipes["blur+resize"]).run_multi(ctx, frames.cpu())
        raise SystemExit("host frames were accepted")
    except ValueError:
        pass
print("py_virtual_failure_injection=ok")
'''


@pytest.mark.gpu
def test_virtual_worlds_from_python():
    out = subprocess.run([sys.executable, "-c", PY_CODE % ROOT], capture_output=True, text=True, timeout=600, env=_env())
    log = out.stdout + out.stderr
    assert out.returncode == 0, log
    for world in (2, 3, 8):
        for pieces in (1, 4, 8):
            assert f"py_virtual_world{world}_{pieces}_pieces=ok" in log, log
    assert "py_virtual_failure_injection=ok" in log, log
