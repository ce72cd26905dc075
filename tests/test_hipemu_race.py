"""TEST INFRASTRUCTURE: the emulator's LDS race detector finds what it should and nothing else (controls).  The kernels of the
library are run under it by `python tests/hipemu/run_race.py` (minutes, not part of this suite)."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_race_detector_controls(tmp_path):
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = "clang++"
    out = str(tmp_path / "librace_controls.so")
    subprocess.check_call([cxx, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-DHIPEMU_RACE", "-fsanitize=thread",
                           "-I", os.path.join(HERE, "hipemu", "include"), os.path.join(HERE, "hipemu", "selftest", "race_controls.cpp"),
                           "-ldl", "-o", out])
    lib = ctypes.CDLL(out)
    scratch = (ctypes.c_int * 128)()
    counts = (ctypes.c_ulong * 4)()
    assert lib.race_controls(scratch, counts) == 0
    fine, same_wave, racy, war = list(counts)
    assert fine == 0 and same_wave == 0, (fine, same_wave)
    assert racy > 0 and war > 0, (racy, war)
