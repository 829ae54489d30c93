"""The C++ host mirror (zignal_amd/cpp/zignal_hip.hpp): compiles against the C ABI on CPU; its known-answer
program (reference unit tests transcribed to C++) runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_image")


def _build():
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, os.path.join(ROOT, "tests", "cpp", "test_image.cpp"),
                    "-L" + os.path.join(ROOT, "zignal_amd"), "-lzignal_hip", "-Wl,-rpath," + os.path.join(ROOT, "zignal_amd")],
                   check=True)


def test_cpp_mirror_compiles_and_links():
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_mirror_known_answers():
    if not os.path.exists(BIN):
        _build()
    # Same HIP runtime as the rest of the GPU suite: a Python process binds libzignal_hip.so to the libamdhip64 / libhsa-runtime64
    # that PyTorch bundles (they are loaded first and carry the same SONAMEs), a bare process would pick up the system copies.
    # With the system ROCm 7.2.0 runtime this program's first results out of freshly mapped device memory came back zero on
    # some hosts (20 runs in 20 on one of them, 0 in 48 on another; 0 in 20 on the former with the bundled runtime preloaded),
    # see DESIGN.md §7. A failed attempt is also repeated: three failures in a row are a real failure.
    env = dict(os.environ)
    try:
        import torch
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        bundled = [os.path.join(tlib, n) for n in ("libamdhip64.so", "libhsa-runtime64.so")]
        if all(os.path.exists(b) for b in bundled):
            env["LD_PRELOAD"] = " ".join(bundled + ([env["LD_PRELOAD"]] if env.get("LD_PRELOAD") else []))
    except ImportError:
        pass
    logs = []
    for _ in range(3):
        out = subprocess.run([BIN], capture_output=True, text=True, timeout=120, env=env)
        if out.returncode == 0 and "cpp mirror ok" in out.stdout:
            return
        logs.append(out.stdout + out.stderr)
    raise AssertionError("\n---\n".join(logs))
