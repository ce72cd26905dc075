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
# -fno-slp-vectorize for EVERY file: hipcc's SLP vectoriser packs adjacent scalar fp32 adds / multiplies into v_pk_add_f32 /
# v_pk_mul_f32 / v_pk_fma_f32.  (1) Correctness: with those in the LayerNorm statistics of k_proj_ln_res, lanes 48..63 of a
# wave lose one accumulate step whenever a workgroup of a DIFFERENT, matrix-heavy kernel (the LSTM or attention kernels,
# launched on a second HIP stream) shares the CU — same binary, bit-exact when it runs alone or next to itself;
# scripts/race_probe.py reproduces it in 11 of 12 launches, the build without packed fp32 in 0 of 20
# (profiles/r03c_packed_fp32_corruption.txt).  The same signature (mean / scale error of whole rows, only with co-resident
# workgroups) was seen in round 1 in k_qkv_proj_ln and worked around there with a full LDS drain.  (2) Performance: packed
# fp32 is not faster on CDNA4 (one v_pk_fma_f32 issues like two v_fma_f32) and the vectoriser undoes the hand-ordered
# MFMA / vector interleave of lh_recur.hip.
NO_SLP = ["-fno-slp-vectorize", "-fno-vectorize"]      # (the loop vectoriser packs fp32 the same way)
# Exception: the direct-form FIR / FFT kernels of the rendering row (not on the separator path) are pure
# fp32 vector arithmetic and run at HALF the rate without v_pk_fma_f32 (0.47 -> 0.93 ms for 256-tap responses); their packed
# chains are plain accumulations without exec-masked side blocks and came through the same stress bit-exact (0 of 120
# launches next to the LSTM / attention kernels, scripts/race_probe.py --call render256|render4096;
# tests/test_render.py::test_render_next_to_lstm_kernels_is_bit_identical keeps checking it).
FILE_FLAGS = {"lh_render.hip": ["-fslp-vectorize", "-fvectorize"]}


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
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result", *NO_SLP, *extra_flags]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        extra_env = os.environ.get("LH_FLAGS_" + src.replace(".hip", "").upper(), "").split()      # bisect hook
        cmd = [hipcc, *flags, *FILE_FLAGS.get(src, []), *extra_env, "-c", os.path.join(CSRC, src), "-o", obj]
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
