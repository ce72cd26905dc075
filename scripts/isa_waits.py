"""Developer tool (no GPU needed): find global loads whose latency the compiled code exposes.

    python scripts/isa_waits.py [--near 25] [file.hip ...]        (default: every kernel source of the library)

Compiles each source for gfx950 (device only, the product build's flags), disassembles it and reports, per kernel,
  * every `s_waitcnt vmcnt(N)` that waits for a LOAD issued fewer than `--near` instructions earlier (the load's latency is
    not covered by anything: a prefetch the scheduler sank next to its use, or a counter the compiler lost behind a branch);
  * rounds of `load -> s_waitcnt vmcnt(0) -> store` (a loop the compiler serialised because it could not prove that the
    store does not alias the next load).
Round 5 found k_emb_qkv (17 serialised rounds per frame and head), k_local_attn (V refill drained / sunk) and the exposed
residual loads of k_emb_convt2 this way (profiles/r05k, r05l).  Prologue / tile-epilogue hits are expected; read the position."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lookoncetohear_amd import build  # noqa: E402



def tools():
    """(hipcc, clang-offload-bundler, llvm-objdump) of the toolchain in use — $HIPCC / PATH / /opt/rocm for hipcc, the llvm
    directory `build._llvm_bin()` resolves for the other two; a missing one is None (callers skip)."""
    import shutil
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    llvm = build._llvm_bin()
    found = [hipcc if os.path.exists(hipcc) or shutil.which(hipcc) else None]
    for t in ("clang-offload-bundler", "llvm-objdump"):
        c = os.path.join(llvm, t)
        found.append(c if os.path.exists(c) else shutil.which(t))
    return tuple(found)


def disassemble(src: str, tmp: str) -> str:
    base = os.path.join(tmp, os.path.basename(src).replace(".hip", ""))
    flags = [f"--offload-arch={build.ARCH}", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=fast", *build.NO_SLP,
             *build.FILE_FLAGS.get(os.path.basename(src), []), "-I", os.path.join(ROOT, "include")]
    hipcc, bundler, objdump = tools()
    subprocess.check_call([hipcc, *flags, "-c", src, "-o", base + ".co"])
    subprocess.check_call([bundler, "--unbundle", "--type=o", f"--input={base}.co",
                           f"--targets=hipv4-amdgcn-amd-amdhsa--{build.ARCH}", f"--output={base}.dev.co"])
    return subprocess.check_output([objdump, "-d", f"--mcpu={build.ARCH}", base + ".dev.co"], text=True)


def kernels(asm: str):
    name, ins = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            if name:
                yield name, ins
            name, ins = m.group(1), []
        elif name and line.strip():
            ins.append(line.strip())
    if name:
        yield name, ins


def scan(ins, near):
    mem, short, serial = [], [], 0
    for i, line in enumerate(ins):
        op = line.split()[0]
        if op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "scratch_")):
            mem.append((i, "load" in op))
        elif op.startswith("s_waitcnt"):
            m = re.search(r"vmcnt\((\d+)\)", line)
            if m and int(m.group(1)) < len(mem):
                j, is_load = mem[len(mem) - 1 - int(m.group(1))]
                if is_load and i - j < near:
                    short.append((i, i - j, sum(1 for x in ins[j:i] if x.startswith("v_mfma"))))
                if int(m.group(1)) == 0 and any(b.startswith(("global_load", "buffer_load")) for b in ins[max(0, i - 8):i]) \
                        and any(a.startswith("global_store") for a in ins[i + 1:i + 13]):
                    serial += 1
    return short, serial


if __name__ == "__main__":
    args = sys.argv[1:]
    near = 25
    if "--near" in args:
        near = int(args[args.index("--near") + 1])
        del args[args.index("--near"):args.index("--near") + 2]
    srcs = args or [os.path.join(build.CSRC, s) for s in build.SOURCES]
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            for name, ins in kernels(disassemble(src, tmp)):
                short, serial = scan(ins, near)
                if short or serial >= 3:
                    print(f"{os.path.basename(src):20s} {name[:60]:60s} {len(ins):5d} instr | serialised load/store rounds: {serial:2d} | "
                          f"waits on a fresh load (pos, instr since issue, MFMAs between): {short[:10]}")
