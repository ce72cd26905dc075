"""Builds the gfx950 C-ABI library in-tree with hipcc (cross-compiles without a GPU).

    python -m lookoncetohear_amd.build [--force]

Output: lookoncetohear_amd/_lookonce_hip.so (git-ignored, but it travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
SOURCES = ["lh_frontend.hip", "lh_lstm.hip", "lh_recur.hip", "lh_pointwise.hip", "lh_attn.hip", "lh_backend.hip", "lh_metrics.hip", "lh_embed.hip", "lh_render.hip", "lh_render_fft.hip", "lh_stream.hip", "lh_comm.hip", "lh_ref32.hip"]
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
# The exact hardware condition, isolated to ONE instruction (scripts/ubench/pk_race*.{hip,py}, profiles/r03c): a packed fp32
# operation (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose op_sel crosses the halves of src1 ONLY — op_sel:[0,1],
# op_sel:[0,1,0] — returns, in lanes 48..63, the result with src1's low-half operand missing (the add returns src0.lo)
# whenever a wave of a matrix-heavy kernel shares the SIMD; default selects, op_sel:[1,0], [1,1], [1,0,0], [0,0,1], the
# neg modifiers and op_sel_hi alone are fine; 8 wait states around every instruction change nothing.
# Exception to NO_SLP: lh_render.hip (direct-form FIR + mixing, SURVEY 8f rank 3, not on the separator path) is pure fp32
# vector arithmetic and runs at HALF the rate without v_pk_fma_f32 (0.47 -> 0.93 ms for 256-tap responses); its packed
# instructions use default selects only.  The FFT kernel (complex multiplies -> crossed selects) is its own file,
# lh_render_fft.hip, built like the rest.  `check_isa` below fails the build if the unsafe form appears anywhere.
FILE_FLAGS = {"lh_render.hip": ["-fslp-vectorize", "-fvectorize"]}


def is_unsafe_packed_fp32(asm_line: str) -> bool:
    """One line of llvm-objdump output: a packed fp32 add / mul / fma whose op_sel crosses the halves of src1 and only src1."""
    import re
    return bool(re.search(r"v_pk_(add|mul|fma)_f32", asm_line) and re.search(r"op_sel:\[0,1(,0)?\]", asm_line))


def _llvm_bin() -> str:
    """Directory of the llvm-objdump that belongs to the hipcc in use: next to $HIPCC, under $ROCM_PATH / $HIP_PATH, or the
    image's /opt/rocm.  A missing tool is an explicit error, not a FileNotFoundError out of subprocess."""
    cands = []
    hipcc = os.environ.get("HIPCC")
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin"))
    for var in ("ROCM_PATH", "HIP_PATH"):
        if os.environ.get(var):
            cands.append(os.path.join(os.environ[var], "lib", "llvm", "bin"))
            cands.append(os.path.join(os.environ[var], "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        # BOTH tools the guard runs: llvm-objdump (disassembly) and llvm-mc (self-test probe) — ADVICE r4: a directory with
        # only the first passed here and guard_selftest then died with a bare FileNotFoundError after the whole compile
        if all(os.path.exists(os.path.join(c, t)) for t in ("llvm-objdump", "llvm-mc")):
            return c
    raise RuntimeError("ISA guard of build.py: llvm-objdump + llvm-mc not found together (looked in " + ", ".join(cands) +
                       "); set ROCM_PATH")


_GUARD_SELFTEST = {}


def guard_selftest() -> None:
    """The guard must recognise the unsafe form in THIS toolchain's disassembly spelling: assemble one v_pk_add_f32 with
    op_sel:[0,1] and one with default selects, disassemble both, and check that exactly the first is flagged.  A newer
    llvm-objdump that prints the modifier differently fails here instead of letting the guard pass silently."""
    llvm = _llvm_bin()
    if llvm in _GUARD_SELFTEST:
        return
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src, obj = os.path.join(td, "g.s"), os.path.join(td, "g.o")
        open(src, "w").write("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\nv_pk_add_f32 v[0:1], v[2:3], v[4:5]\n"
                             "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]\n")
        r = subprocess.run([os.path.join(llvm, "llvm-mc"), "-arch=amdgcn", f"-mcpu={ARCH}", "-filetype=obj", src, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("ISA guard self-test: llvm-mc could not assemble the probe: " + r.stderr[-300:])
        d = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", f"--mcpu={ARCH}", obj], capture_output=True, text=True).stdout
    lines = [ln for ln in d.splitlines() if "v_pk_" in ln]
    flags = [is_unsafe_packed_fp32(ln) for ln in lines]
    if flags != [True, False, True]:
        raise RuntimeError(f"ISA guard self-test failed: expected [unsafe, safe, unsafe], got {flags} for {lines}")
    _GUARD_SELFTEST[llvm] = True


def unsafe_packed_fp32(lib_path: str):
    """Disassembles the gfx950 code object inside `lib_path` and returns the packed-fp32 instructions whose op_sel crosses
    the halves of src1 only (see above)."""
    import re
    import tempfile
    llvm = _llvm_bin()
    with tempfile.TemporaryDirectory() as td:
        # newer llvm-objdump finds the fat-binary section itself — and writes every bundle it extracts next to its input:
        # work on a copy in the temporary directory
        tmp_lib = os.path.join(td, os.path.basename(lib_path))
        shutil.copyfile(lib_path, tmp_lib)
        out = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--offloading", tmp_lib], capture_output=True, text=True,
                             cwd=td)
        text = out.stdout
        if "v_mfma" not in text:                        # older llvm-objdump: extract the code object by hand
            raw = open(lib_path, "rb").read()
            i = raw.find(b"\x7fELF", 1)
            hits = []
            while i > 0:
                hits.append(i)
                i = raw.find(b"\x7fELF", i + 1)
            text = ""
            for k, off in enumerate(hits):
                co = os.path.join(td, f"co{k}.o")
                open(co, "wb").write(raw[off:])
                r = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", f"--mcpu={ARCH}", co], capture_output=True, text=True)
                if "v_mfma" in r.stdout or "v_pk_" in r.stdout:
                    text += r.stdout
    bad = [line.strip() for line in text.splitlines() if is_unsafe_packed_fp32(line)]
    return bad, text.count("v_pk_fma_f32") + text.count("v_pk_add_f32") + text.count("v_pk_mul_f32"), ("v_mfma" in text)


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build_hip(force: bool = False, verbose: bool = True, extra_flags=(), out: str = LIB) -> str:
    """`extra_flags` / `out`: A/B variants of the same sources (e.g. -DLH_ACT_MERGED=1) built next to the product
    library and selected with LOOKONCE_HIP_LIB (scripts/gpu_ab.sh); the product build uses the defaults."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "lh_common.h"), os.path.join(CSRC, "lh_split.h"), os.path.join(CSRC, "lh_quad.h"),
                                                       os.path.join(os.path.dirname(PKG), "include", "lookonce_hip.h")]
    if not force and _newer(out, deps):
        return out
    guard_selftest()                         # before the compile: a toolchain the guard cannot read fails in a second, not after minutes
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
    # link under a temporary name: a library that fails the ISA guard must never be loadable
    tmp_out = out + ".unchecked"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp_out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    bad, n_packed, seen = unsafe_packed_fp32(tmp_out)
    if not seen or bad:
        rejected = out + ".rejected"
        os.replace(tmp_out, rejected)
        if os.path.exists(out):
            os.remove(out)                   # the previous build no longer matches the sources either
        if not seen:
            raise RuntimeError(f"{rejected}: could not disassemble the gfx950 code object (ISA guard of build.py)")
        raise RuntimeError(f"{rejected}: {len(bad)} packed-fp32 instructions with src1-crossed op_sel (unsafe on gfx950 next to "
                           f"matrix-heavy kernels, see build.py), e.g. {bad[0]}")
    os.replace(tmp_out, out)
    if verbose:
        print(f"ISA guard: {n_packed} packed fp32 instructions, none with src1-crossed op_sel", flush=True)
    return out


if __name__ == "__main__":
    # python -m lookoncetohear_amd.build [--force] [--variant NAME -DX=1 ...]  -> _lookonce_hip_NAME.so
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build_hip(force=True, extra_flags=[a for a in sys.argv if a.startswith(("-D", "-f", "-m"))],
                        out=os.path.join(PKG, f"_lookonce_hip_{name}.so")))
    else:
        print(build_hip(force="--force" in sys.argv))
