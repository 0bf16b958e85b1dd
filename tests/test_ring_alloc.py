"""tests/fake_rccl/ring_alloc.hpp — the allocator of the shared-memory RCCL stand-in's data segments (test infrastructure the
multi-process transport tests rest on) — driven on the HOST over 800 000 random steps: regions in use never overlap, never leave
the segment, are released oldest first, and a request that fits the segment succeeds once enough has been released."""
import os
import subprocess

from helpers import ROOT


def test_ring_allocator(tmp_path):
    exe = tmp_path / "ring_alloc_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("FPIC_NATIVE_CXXFLAGS", "").split(), os.path.join(ROOT, "tests", "native", "ring_alloc_test.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], timeout=120).decode()
    assert out.strip().splitlines()[-1] == "ok", out
    assert int(out.split("wraps=")[1].split()[0]) > 1000
