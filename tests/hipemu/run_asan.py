"""TEST INFRASTRUCTURE: the emulator runs of the kernel sources under AddressSanitizer.

    python tests/hipemu/run_asan.py [-k EXPR]

Builds tests/hipemu/_build/liblookonce_emu_asan.so (the unmodified kernel sources as host C++, -fsanitize=address) and runs
tests/test_emu_kernels.py against it with the ASan runtime preloaded into the interpreter: every global-memory access of a kernel
is checked against the torch allocation it points into, every LDS access (`__shared__` = static array here) against its array.
Round 3: the 21 emulator tests are clean; a deliberately short state buffer handed to lh_ring_unpack is reported as
heap-buffer-overflow in k_ring_unpack (negative control).  `--ubsan` builds with -fsanitize=undefined,alignment instead
(misaligned vector accesses, signed overflow, out-of-range shifts): also clean.  Not part of `pytest -m "not gpu"` (needs
LD_PRELOAD)."""
import glob
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
be.build_emu = lambda *a, **k: %(lib)r          # the fixtures load the sanitised library
import pytest
sys.exit(pytest.main(["-x", "-q", os.path.join(%(root)r, "tests", "test_emu_kernels.py"), os.path.join(%(root)r, "tests", "test_render.py"), "-m", "not gpu", "-p", "no:cacheprovider"] + sys.argv[1:]))
"""


def main():
    from tests.hipemu.build_emu import build_emu
    args = sys.argv[1:]
    ubsan = "--ubsan" in args
    args = [a for a in args if a != "--ubsan"]
    if ubsan:
        lib = os.path.join(HERE, "_build", "liblookonce_emu_ubsan.so")
        flags = ["-fsanitize=undefined,alignment", "-fno-sanitize=vptr,function", "-fno-omit-frame-pointer", "-shared-libsan"]
        rt_name, env_extra = "libclang_rt.ubsan_standalone-x86_64.so", {"UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}
    else:
        lib = os.path.join(HERE, "_build", "liblookonce_emu_asan.so")
        flags = ["-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan"]
        rt_name = "libclang_rt.asan-x86_64.so"
        env_extra = {"ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:verify_asan_link_order=0"}
    build_emu(force=False, extra_flags=flags, out=lib)
    rt = sorted(glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/" + rt_name))
    if not rt:
        sys.exit("no %s under /opt/rocm" % rt_name)
    env = dict(os.environ, LD_PRELOAD=rt[-1], **env_extra)
    return subprocess.call([sys.executable, "-c", CHILD % {"root": ROOT, "lib": lib}] + args, env=env)


if __name__ == "__main__":
    sys.exit(main())
