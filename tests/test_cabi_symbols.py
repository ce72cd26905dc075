"""CPU: the hipcc-built gfx950 library loads and exports every symbol include/lookonce_hip.h declares (no
compute calls: there is no GPU here); the ctypes signature table matches the header one to one; the drop-in
class mirrors the reference constructor / state-dict surface; the product path refuses CPU tensors."""
import os
import re

import pytest
import torch

from lookoncetohear_amd import _cabi
from lookoncetohear_amd.net import Net
from oracle import tfgridnet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lookonce_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef LH_LEGACY.*?#endif", "", src, flags=re.S)       # lab-build entry points: not in the product library
    out = {}
    for m in re.finditer(r"\bint\s+(lh_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = args
    return out


def test_header_matches_ctypes_table():
    hdr = _header_functions()
    assert set(hdr) == set(_cabi.SIGNATURES), set(hdr) ^ set(_cabi.SIGNATURES)
    for name, args in hdr.items():
        sig = _cabi.SIGNATURES[name]
        assert len(args) == len(sig), name
        for a, t in zip(args, sig):
            is_ptr = "*" in a or a.startswith("lh_stream_t")
            assert is_ptr == (t is _cabi.c_void_p), (name, a)


def test_hip_library_builds_loads_and_exports():
    from lookoncetohear_amd.build import build_hip
    lib = _cabi.Lib(build_hip())
    for name in _header_functions():
        assert lib.raw(name) is not None
    # superseded kernels are not in the product library (they build with -DLH_LEGACY for the A/B lab
    # and the emulator): their entry point is absent and their tuning switch is refused
    import ctypes
    assert not hasattr(ctypes.CDLL(lib.path), "lh_emb_axis")
    assert lib.raw("lh_set_tuning")(2, 2) == 2 and lib.raw("lh_set_tuning")(2, 0) == 0
    for v in (0, 1, 2):                                   # the attention GEMM's measured-slower pipelines: lab builds only
        assert lib.raw("lh_set_tuning")(17, v) == 2
    assert lib.raw("lh_set_tuning")(17, 3) == 0
    assert lib.raw("lh_abi_version")() == _cabi.ABI_VERSION
    assert lib.raw("lh_check_config")(192, 128, 2, 64, 3, 64, 4, 50, 2, 256) == 0


def test_library_has_no_unsafe_packed_fp32():
    """ISA guard of build.py: a packed fp32 instruction whose op_sel crosses the halves of src1 only (op_sel:[0,1] /
    [0,1,0]) returns wrong lanes 48..63 on gfx950 whenever a wave of a matrix-heavy kernel shares its SIMD
    (profiles/r03c_packed_fp32_corruption.txt, scripts/ubench/pk_race.*).  The shipped code object must not contain one."""
    from lookoncetohear_amd.build import build_hip, unsafe_packed_fp32
    bad, n_packed, seen = unsafe_packed_fp32(build_hip())
    assert seen, "could not disassemble the gfx950 code object"
    assert not bad, bad[:3]
    assert n_packed < 200            # the direct-form FIR kernel and the quad-lane LSTM step keep (default-select) packed fp32


def test_isa_guard_classification():
    """The forms measured safe / unsafe by scripts/ubench/pk_race* (profiles/r03c_packed_fp32_corruption.txt), as llvm-objdump
    prints them — including the one hipcc formed from two horizontal sums in lh_stream.hip (caught by the guard, round 3)."""
    from lookoncetohear_amd.build import is_unsafe_packed_fp32 as bad
    assert bad("v_pk_add_f32 v[74:75], v[76:77], v[76:77] op_sel:[0,1] op_sel_hi:[1,0]// 000000002C80: D3B2504A 0802994C")
    assert bad("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]")
    assert bad("v_pk_mul_f32 v[8:9], v[2:3], v[4:5] op_sel:[0,1]")
    for ok in ("v_pk_fma_f32 v[90:91], v[14:15], v[74:75], 0 op_sel_hi:[1,1,0]",
               "v_pk_fma_f32 v[76:77], v[10:11], v[78:79], v[90:91]",
               "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]",
               "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]",
               "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]",
               "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]",
               "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]",
               "v_pk_add_f16 v0, v1, v2 op_sel:[0,1]",                      # packed fp16: not the affected unit
               "v_fma_f32 v0, v1, v2, v3"):
        assert not bad(ok), ok


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _cabi.Lib(str(tmp_path / "nope.so"))


def test_dropin_surface():
    cfg = O.Cfg(**O.TSH_PARAMS)
    net = Net(**O.TSH_PARAMS)
    man = O.param_manifest(cfg)
    sd = net.state_dict()
    assert set(sd) == set(man)
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in man.items())
    assert sum(p.numel() for p in net.parameters()) == 2037960
    st = net.init_buffers(3, "cpu")
    ref = O.init_state(cfg, 3)
    assert {k: tuple(v.shape) for k, v in O.flat_state(st).items()} == {k: tuple(v.shape) for k, v in O.flat_state(ref).items()}
    with pytest.raises(RuntimeError, match="no CPU"):
        net.eval()(torch.zeros(1, 2, 1000), torch.zeros(1, 1, 256))
    with pytest.raises(NotImplementedError):
        Net()                                                           # reference defaults: unsupported shapes


def test_exchange_entry_points_validate_arguments():
    """lh_comm_* / lh_allreduce_f64 (the sharded eval's one exchange step for Python-free hosts): exported, and bad
    arguments are refused before RCCL is touched (no GPU here, so no communicator is created)."""
    from lookoncetohear_amd.build import build_hip
    lib = _cabi.Lib(build_hip())
    assert lib.raw("lh_comm_unique_id")(None) == 1                      # LH_ERR_ARG
    assert lib.raw("lh_comm_init")(None, 2, 0, None) == 1
    assert lib.raw("lh_allreduce_f64")(None, None, 4, None) == 1
    assert lib.raw("lh_comm_destroy")(None) == 1
