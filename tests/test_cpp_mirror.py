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
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpp mirror ok" in out.stdout
