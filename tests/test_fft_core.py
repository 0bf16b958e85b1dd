"""The arithmetic core of the library's own FFT passes (fusion-sim_amd/csrc/fes_fft.hpp: radix-2/4/8 butterflies and
the Stockham index maps the device kernels wrap) built for the HOST with g++ and checked against a direct O(N^2) transform
in long double: every supported length 2 .. 1024, both directions, float and double.  No GPU involved; the kernels
themselves are held to the oracle's Poisson solve in tests/test_gpu_es3d.py."""
import os
import subprocess

from helpers import ROOT


def test_fft_core_matches_a_direct_transform(tmp_path):
    exe = tmp_path / "fft_core_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("FPIC_NATIVE_CXXFLAGS", "").split(), os.path.join(ROOT, "tests", "native", "fft_core_test.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], timeout=120).decode()
    assert out.strip().splitlines()[-1] == "ok", out
    assert out.count("N=") == 10
