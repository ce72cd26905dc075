"""TEST INFRASTRUCTURE: the emulator runs of the kernel sources under the LDS race detector.

    python tests/hipemu/run_race.py [-k EXPR]

Builds tests/hipemu/_build/liblookonce_emu_race.so — the unmodified kernel sources as host C++ with -DHIPEMU_RACE
-fsanitize=thread, WITHOUT the ThreadSanitizer runtime: the instrumentation's per-access calls land in hooks of the emulator
header — and runs tests/test_emu_kernels.py against it.  Model (hip_runtime.h): two accesses to the same LDS byte by different
WAVES of a workgroup, at least one a write, with no workgroup barrier between them, are a race; lanes of one wave are in
lockstep.  Every distinct (code address, kind) is printed once as "hipemu RACE ..." with the kernel's name; the exit code
is non-zero if anything was reported.  Controls: tests/test_hipemu_race.py.  Minutes, not part of `pytest -m "not gpu"`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CHILD = """
import os, sys
sys.path.insert(0, %(root)r)
import tests.hipemu.build_emu as be
be.build_emu = lambda *a, **k: %(lib)r          # the fixtures load the instrumented library
import pytest
rc = pytest.main(["-q", os.path.join(%(root)r, "tests", "test_emu_kernels.py"), os.path.join(%(root)r, "tests", "test_render.py"), "-m", "not gpu", "-p", "no:cacheprovider"] + sys.argv[1:])
import ctypes
n = ctypes.CDLL(%(lib)r).hipemu_race_reports
n.restype = ctypes.c_ulong
print("hipemu race detector: %%d conflicting accesses reported" %% n(), flush=True)
sys.exit(int(rc) or (1 if n() else 0))
"""


def main():
    from tests.hipemu.build_emu import build_emu
    lib = os.path.join(HERE, "_build", "liblookonce_emu_race.so")
    build_emu(force=False, extra_flags=["-DHIPEMU_RACE", "-fsanitize=thread", "-fno-omit-frame-pointer"], out=lib)
    return subprocess.call([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib}] + sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())
