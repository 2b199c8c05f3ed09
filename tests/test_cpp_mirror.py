"""Builds and runs tests/cpp/test_mirror.cpp: the C++ host-side mirror of the reference's API
(include/ws_watershed.hpp) over the C ABI."""
import os
import subprocess

import pytest

import __graft_entry__ as ge
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_mirror")


def _build():
    ge.build_hip()
    ol.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
    deps = [src, os.path.join(ROOT, "include", "ws_watershed.hpp"), os.path.join(ROOT, "include", "ws_hip.h")]
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", BIN, src,
                           "-L" + os.path.join(ROOT, "rustronomy-watershed_amd"), "-lws_hip",
                           "-L" + os.path.join(ROOT, "oracle", "_build"), "-lws_oracle",
                           "-Wl,-rpath,$ORIGIN/../../../rustronomy-watershed_amd",
                           "-Wl,-rpath,$ORIGIN/../../../oracle/_build"])
    return BIN


def test_cpp_mirror_cpu():
    out = subprocess.run([_build(), "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cpu checks ok" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_gpu():
    out = subprocess.run([_build(), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gpu checks ok" in out.stdout
    assert "shim sequence ok" in out.stdout          # the Rust shim's ABI call order, executed from C++
    assert "group sequence ok" in out.stdout         # three ranks behind ws_segment_tiled (local group), vs the oracle
