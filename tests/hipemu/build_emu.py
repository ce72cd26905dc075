"""TEST INFRASTRUCTURE: compiles the unmodified kernel sources as host C++ against tests/hipemu/include
(see that header).  Output: tests/hipemu/_build/liblookonce_emu.so — used by `-m "not gpu"` tests only.  Built with
-DLH_LEGACY: the superseded kernels the product library no longer contains stay under test here as independent
cross-checks of their successors."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lookoncetohear_amd", "csrc")
OUT = os.path.join(HERE, "_build", "liblookonce_emu.so")
SOURCES = ["lh_frontend.hip", "lh_lstm.hip", "lh_recur.hip", "lh_pointwise.hip", "lh_attn.hip", "lh_backend.hip", "lh_metrics.hip", "lh_embed.hip", "lh_render.hip", "lh_render_fft.hip", "lh_stream.hip", "lh_comm.hip", "lh_ref32.hip"]


def build_emu(force=False, extra_flags=(), out=OUT):
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = "clang++"
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "lh_common.h"), os.path.join(CSRC, "lh_split.h"), os.path.join(CSRC, "lh_quad.h"),
                                                       os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                                                       os.path.join(ROOT, "include", "lookonce_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-march=native", "-DLH_LEGACY", *extra_flags,
           "-I", os.path.join(HERE, "include"), *[os.path.join(CSRC, s) for s in SOURCES], "-ldl", "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_emu(force=True))
