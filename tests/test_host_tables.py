"""csrc/fpic_host.cpp — the reference's host-side tables (empic.js:44-46 constants, toFixed(20) shader literals, the 11x11
stamp of empic.js:949-971) — built for the HOST with g++ and checked against the numbers SURVEY.md 8(c) records from the
reference's own JavaScript.  No GPU.  `make -C fusion-sim_amd sanitize` runs it under AddressSanitizer + UBSan."""
import os
import subprocess

from helpers import ROOT


def test_host_tables(tmp_path):
    exe = tmp_path / "host_tables_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("FPIC_NATIVE_CXXFLAGS", "").split(), os.path.join(ROOT, "tests", "native", "host_tables_test.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], timeout=60).decode()
    assert out.strip().splitlines()[-1] == "ok", out
