"""The integer rules by which a work item of the tiled pushes takes its slots in groups of PPT
(fusion-sim_amd/csrc/fes_groups.hpp: pieces of a tile in a joint work list, the earlier-item rule, the one group a two-part
launch hands from the interior to the upper face), built for the HOST with g++ and driven over 40 000 random bin tables:
every group of every species is pushed exactly once — by one launch, or by the two parts together — and the interior's
launch never pushes a particle of a face layer (whose deposits must be complete before the ghost planes travel).  Found
while writing it: an interior whose last group holds no face particle (a species absent from the upper face) lost that
group, and an interior of fewer than four slots had its neighbours' shared group pushed twice.  The same program checks
held_plane() — the place of a plane in a rank's slab-only arrays, first and last rank's wrapping halo included — and the
window's periodic distance wrap_near()."""
import os
import subprocess

from helpers import ROOT


def test_group_rules_partition_every_bin_table(tmp_path):
    exe = tmp_path / "groups_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("FPIC_NATIVE_CXXFLAGS", "").split(), os.path.join(ROOT, "tests", "native", "groups_test.cpp"), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], timeout=120).decode()
    assert out.strip().splitlines()[-1] == "ok", out
    assert int(out.split("cases=")[1].split()[0]) > 30000
