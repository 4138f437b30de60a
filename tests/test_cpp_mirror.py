"""Builds tests/cpp/test_cpp_mirror.cpp (the C++ mirror of the reference API in include/alice_codec.hpp) with g++
against libalice_codec.so and runs it: host-only checks on CPU, the full walk-through under -m gpu."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run():
    exe = os.path.join(tempfile.mkdtemp(prefix="alice_cpp_"), "test_cpp_mirror")
    libdir = os.path.join(ROOT, "alice-codec_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_cpp_mirror.cpp"), "-L", libdir, "-lalice_codec",
                           "-Wl,-rpath," + libdir, "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_cpp_mirror_host_checks(codec):
    out = _build_and_run()
    assert "CPP MIRROR OK" in out


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(gpu_codec):
    out = _build_and_run()
    assert out.strip() == "CPP MIRROR OK"
