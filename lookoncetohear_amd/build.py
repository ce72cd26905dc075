"""Builds the gfx950 C-ABI library in-tree with hipcc (cross-compiles without a GPU).

    python -m lookoncetohear_amd.build [--force]

Output: lookoncetohear_amd/_lookonce_hip.so (git-ignored, but it travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SOURCES = ["lh_frontend.hip", "lh_lstm.hip", "lh_recur.hip", "lh_pointwise.hip", "lh_attn.hip", "lh_backend.hip", "lh_metrics.hip", "lh_embed.hip", "lh_render.hip", "lh_stream.hip", "lh_comm.hip"]
LIB = os.path.join(PKG, "_lookonce_hip.so")
ARCH = "gfx950"
# per-file flags.  lh_recur.hip: its recurrent steps are hand-ordered (one MFMA, then the vector instructions that fit in its
# shadow, then a scheduling fence); the SLP vectoriser would gather the scalar fp32 operations of different slots into
# packed instructions at one place and undo that order.
FILE_FLAGS = {"lh_recur.hip": ["-fno-slp-vectorize"]}


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build_hip(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB) -> str:
    """`extra_flags` / `out`: A/B variants of the same sources (e.g. -DLH_ACT_MERGED=1) built next to the product
    library and selected with LOOKONCE_HIP_LIB (scripts/gpu_ab.sh); the product build uses the defaults."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "lh_common.h"), os.path.join(CSRC, "lh_split.h"),
                                                       os.path.join(os.path.dirname(PKG), "include", "lookonce_hip.h")]
    if not force and _newer(out, deps):
        return out
    objdir = os.path.join(PKG, "build" if out == LIB else "build_" + os.path.basename(out).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", *extra_flags]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # python -m lookoncetohear_amd.build [--force] [--variant NAME -DX=1 ...]  -> _lookonce_hip_NAME.so
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build_hip(force=True, extra_flags=[a for a in sys.argv if a.startswith(("-D", "-f", "-m"))],
                        out=os.path.join(PKG, f"_lookonce_hip_{name}.so")))
    else:
        print(build_hip(force="--force" in sys.argv))
