"""The C++ host mirror (zignal_amd/cpp/zignal_hip.hpp): compiles against the C ABI on CPU; its programs run on the GPU as
bare processes — the configuration a Zig or C++ caller has: the image's system HIP runtime, no PyTorch in the process.

  tests/cpp/test_image.cpp         known answers of the reference's own unit tests through Image<T> (host pointers)
  tests/cpp/test_device_image.cpp  DeviceImage<T>: config 5's [blur, resize] resident in HBM against the oracle, device-event
                                   timings, the banded host-pointer pipeline, graph capture of ops that take scratch
"""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "test_image")
BIN_DEV = os.path.join(CPP, "test_device_image")


def _build():
    lib_dir, oracle_dir = os.path.join(ROOT, "zignal_amd"), os.path.join(ROOT, "oracle")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, os.path.join(CPP, "test_image.cpp"),
                    "-L" + lib_dir, "-lzignal_hip", "-Wl,-rpath," + lib_dir], check=True)
    # the oracle is linked into the TEST program as its checker (never into the product)
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN_DEV, os.path.join(CPP, "test_device_image.cpp"),
                    "-L" + lib_dir, "-lzignal_hip", "-L" + oracle_dir, "-l:liboracle.so", "-pthread",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + oracle_dir], check=True)


def _ensure_built():
    if not (os.path.exists(BIN) and os.path.exists(BIN_DEV)):
        _build()


def test_cpp_mirror_compiles_and_links():
    _build()
    assert os.path.exists(BIN) and os.path.exists(BIN_DEV)


def _run(args, timeout=300):
    out = subprocess.run(args, capture_output=True, text=True, timeout=timeout)
    return out.returncode, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_known_answers():
    _ensure_built()
    rc, log = _run([BIN], 120)  # one attempt: a failure is a failure
    assert rc == 0 and "cpp mirror ok" in log, log


@pytest.mark.gpu
def test_device_image_chain_banded_host_layer_and_graph_capture():
    """Full sizes: 1080p config-5 chain and 4096^2 Rgba(f32) blur, both bit-equal to the oracle inside the program."""
    import torch

    import zignal_amd as zg
    from oracle import pyoracle as oracle

    _ensure_built()
    rc, log = _run([BIN_DEV], 600)
    assert rc == 0 and "device image ok" in log, log
    vals = {k: float(v) for k, v in re.findall(r"^(\w+)=([0-9.]+)(?: [0-9. ]+)?$", log, re.M)}
    print(log)

    # the same resident chain driven from Python (what bench.py's legs time), with device events on the same stream
    src = zg.Image(torch.from_numpy(oracle.synth_u8(3, (1080, 1920, 4))).cuda())
    mid = zg.Image(torch.empty((1080, 1920, 4), dtype=torch.uint8, device="cuda"))
    dst = zg.Image(torch.empty((540, 960, 4), dtype=torch.uint8, device="cuda"))

    def chain():
        src.gaussian_blur(0.6, out=mid)
        mid.resize(dst, zg.Interpolation.bilinear)

    for _ in range(20):
        chain()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        chain()
    b.record()
    torch.cuda.synchronize()
    py_us = a.elapsed_time(b) * 1000.0 / 200
    cpp_us = vals["chain_1080p_rgba8_blur_resize_us"]
    print(f"resident [blur, resize] on one 1080p Rgba(u8) frame: C++ DeviceImage {cpp_us:.1f} us, Python mirror {py_us:.1f} us")
    assert cpp_us <= 1.10 * py_us + 2.0, (cpp_us, py_us)
    # the compiled-language mirror reaches the measured kernel rate (bench.py's headline: ~0.09 ms per 4096^2 Rgba(f32) frame)
    assert vals["resident_blur_rgba_f32_4096_us"] < 120.0, vals
    # and the host-pointer layer overlaps its two PCIe trips
    print(f"host-pointer gaussianBlur 4096^2 Rgba(f32): banded {vals['host_blur_rgba_f32_4096_banded_ms']:.2f} ms, "
          f"whole-frame {vals['host_blur_rgba_f32_4096_whole_ms']:.2f} ms")
    assert vals["host_blur_rgba_f32_4096_banded_ms"] < vals["host_blur_rgba_f32_4096_whole_ms"], vals


@pytest.mark.gpu
def test_bare_process_is_stable_over_twenty_runs():
    """The system-runtime configuration used to need a retry loop (scratch from the stream-ordered pool came back zeroed);
    with the caching allocator, graph capture included, twenty consecutive bare-process runs pass, none repeated."""
    _ensure_built()
    for i in range(20):
        rc, log = _run([BIN_DEV, "quick"], 120)
        assert rc == 0 and "device image ok" in log, f"run {i}:\n{log}"


@pytest.mark.gpu
def test_zg_multi_world_gt_1_runs_wherever_two_gpus_are_visible():
    """zg_multi's world > 1 branches (shard ownership, staging on the owners, both communicators, 1 and 4 pieces per shard) against the
    one-device result, from a bare C++ process. Needs two GPUs; the one-GPU boxes of the test tier skip it — visibly, by this test's
    own decision rather than the program's."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the world > 1 branches of zg_multi need two")
    _ensure_built()
    rc, log = _run([BIN_DEV], 900)
    assert rc == 0 and "device image ok" in log, log
    assert "multi_world_gt_1=skipped" not in log, log
    world = torch.cuda.device_count()
    for pieces in ("1", "4"):
        assert re.search(rf"^multi_{world}gpu_{pieces}_pieces_ms=", log, re.M), log
