"""ctypes binding of the C-ABI library declared in include/lookonce_hip.h.

The product path loads ONLY the hipcc-built gfx950 library `lookoncetohear_amd/_lookonce_hip.so` and fails
loudly when it is missing: there is no CPU fallback (the CPU oracle under `oracle/` is test infrastructure).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_PKG, "_lookonce_hip.so")
ABI_VERSION = 14

_P, _I = c_void_p, c_int
# name -> argtypes; mirrors include/lookonce_hip.h one to one (tests/test_cabi_symbols.py checks both ways)
SIGNATURES = {
    "lh_abi_version": [],
    "lh_check_config": [_I] * 10,
    "lh_stft_conv_in": [_P] * 7 + [_I] * 3 + [_P],
    "lh_embed_proj_ln": [_P] * 7 + [_I, _P],
    "lh_set_tuning": [_I, _I],
    "lh_ln_lstm_intra": [_P] * 6 + [_I, _I, _P],
    "lh_ln_lstm_inter": [_P] * 10 + [_I, _I, _I, _P],
    "lh_intra_stream": [_P] * 5 + [_I, _P],
    "lh_inter_matvec": [_P] * 11 + [_I, _I, _P],
    "lh_intra_block": [_P] * 6 + [_I, _P],
    "lh_inter_block": [_P] * 10 + [_I, _I, _P],
    "lh_linear_res": [_P] * 5 + [_I, _I, _P],
    "lh_qkv_proj_ln": [_P] * 14 + [_I, _I, _P],
    "lh_local_attn": [_P] * 4 + [_I, _I, _P],
    "lh_ring_pack": [_P] * 4 + [_I, _I, _P],
    "lh_ring_unpack": [_P] * 4 + [_I, _I, _P],
    "lh_ring_advance": [_P, _I, _P],
    "lh_proj_ln_res": [_P] * 9 + [_I, _I, _P],
    "lh_deconv_istft": [_P] * 10 + [_I, _I, _I, _P],
    # time windows (ABI 14): the five block stages on frames [t0, t0 + Tc) of [B][T][97][64] buffers (net.py `time_chunks`)
    "lh_intra_block_win": [_P] * 6 + [_I, _I, _I, _I, _P],
    "lh_inter_block_win": [_P] * 10 + [_I, _I, _I, _I, _I, _P],
    "lh_ln_lstm_intra_win": [_P] * 6 + [_I, _I, _I, _I, _P],
    "lh_linear_res_win": [_P] * 5 + [_I, _I, _I, _I, _I, _P],
    "lh_inter_matvec_win": [_P] * 11 + [_I, _I, _I, _I, _I, _P],
    "lh_qkv_proj_ln_win": [_P] * 14 + [_I, _I, _I, _I, _P],
    "lh_local_attn_win": [_P] * 4 + [_I, _I, _I, _I, _P],
    "lh_proj_ln_res_win": [_P] * 9 + [_I, _I, _I, _I, _P],
    "lh_emb_frontend": [_P] * 10 + [_I, _I, _I, _P],
    "lh_emb_axis_fused": [_P] * 8 + [_I, _I, _I, _I, _I, _P],
    "lh_emb_axis_mv": [_P] * 9 + [_I, _I, _I, _I, _P],
    "lh_emb_attn_block": [_P] * 24 + [_I, _I, _P],
    "lh_emb_head": [_P] * 7 + [_I, _I, _P],
    "lh_render_binaural": [_P] * 8 + [_I, _I, _I, _I, _P],
    "lh_metric_sums": [_P] * 8 + [_I, _I, _I, _P],
    "lh_range_status": [_P, _P],
    "lh_range_flag_copy": [_P, _P, _P],
    "lh_range_flag_clear": [_P, _P],
    "lh_selftest_fp16_subnormal": [_P],
    "lh_comm_unique_id": [_P],
    "lh_comm_init": [_P, _I, _I, _P],
    "lh_allreduce_f64": [_P, _P, _I, _P],
    "lh_comm_destroy": [_P],
    # plain-fp32 reference kernels of the frame stages (gemm_mode "f32all", lh_ref32.hip)
    "lh_ref32_stft_conv_in": [_P] * 8 + [_I, _I, _I, _P],
    "lh_ref32_linear": [_P] * 6 + [_I, _I, _I, _P],
    "lh_ref32_head_ln": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "lh_ref32_local_attn": [_P] * 4 + [_I, _I, _P],
    "lh_ref32_proj_ln_res": [_P] * 11 + [_I, _I, _P],
    "lh_ref32_deconv_istft": [_P] * 10 + [_I, _I, _P],
}
# entry points of lab / emulator builds only (-DLH_LEGACY, include/lookonce_hip.h `#ifdef LH_LEGACY`): bound when present
LEGACY_SIGNATURES = {
    "lh_emb_axis": [_P] * 10 + [_I, _I, _I, _P],
}
ERRORS = {1: "LH_ERR_ARG", 2: "LH_ERR_UNSUPPORTED", 3: "LH_ERR_LAUNCH", 4: "LH_ERR_RANGE"}


class Lib:
    """A loaded C-ABI library with typed entry points; every call asserts the 0 = OK convention."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: the HIP extension is not built. Run `python -m lookoncetohear_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        self.path = path
        # Import torch BEFORE dlopen: PyTorch-ROCm bundles its own libamdhip64.so, and the library must bind to the
        # same HIP runtime that owns torch's streams and allocations.  Loaded first, it would pull in /opt/rocm's
        # runtime as a second copy and every launch on a torch stream would fail (LH_ERR_LAUNCH).
        import torch  # noqa: F401
        self._dll = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self._dll, name)          # AttributeError here = missing export
            fn.argtypes = argtypes
            fn.restype = c_int
        for name, argtypes in LEGACY_SIGNATURES.items():      # lab / emulator builds (-DLH_LEGACY) only
            fn = getattr(self._dll, name, None)
            if fn is not None:
                fn.argtypes = argtypes
                fn.restype = c_int
        v = self._dll.lh_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {v}, expected {ABI_VERSION}")

    def call(self, name: str, *args) -> None:
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed: {ERRORS.get(rc, rc)}")

    def raw(self, name: str):
        return getattr(self._dll, name)


def device_of(t):
    """Context manager: make `t`'s GPU the current HIP device around raw C-ABI launches (they go to the CURRENT device,
    and the reference eval driver builds `cuda:N` tensors without torch.cuda.set_device, src/ts_hear_test.py:175).
    """
    import torch
    return torch.cuda.device(t.device)


_hip_lib = None
_selftested = set()


class HipHost:
    """Device plumbing of the host classes (`Net`, `EmbedTFGridNet`, `BinauralRenderer`): where the C-ABI library comes
    from, which HIP stream the launches go to, which device is current around them, and where a caller-owned flag word
    lives.  ROCm device tensors only — there is no CPU path.  (tests/hipemu subclasses the hosts and overrides exactly
    these four methods to drive the same host code over the emulated library; the product classes carry no test hook.)"""
    _host_name = "this module"

    def _lib(self, t) -> "Lib":
        if not t.is_cuda:
            raise RuntimeError(f"lookoncetohear_amd.{self._host_name} runs on an MI355X (ROCm device tensors); there is no "
                               "CPU path. Move the module and its inputs to cuda.")
        lib = load()
        import torch
        selftest_device(lib, t.device.index if t.device.index is not None else torch.cuda.current_device())
        return lib

    def _stream(self, device) -> int:
        import torch
        return torch.cuda.current_stream(device).cuda_stream

    def _device_ctx(self, t):
        return device_of(t)

    def _flag_words(self, device):
        """Two zeroed 32-bit words in pinned host memory: device-accessible, and readable by the host without a copy."""
        import torch
        return torch.zeros(2, dtype=torch.int32).pin_memory()


def selftest_device(lib: "Lib", device_index: int) -> None:
    """Once per process and device: the un-rescaled split relies on the matrix core taking fp16 SUBNORMAL operands at full
    value (lh_selftest_fp16_subnormal).  A build / mode that flushes them would silently lose the lo halves of every value
    below 2^-3 — refuse to run instead."""
    if device_index in _selftested:
        return
    import torch
    with torch.cuda.device(device_index):
        rc = lib.raw("lh_selftest_fp16_subnormal")(torch.cuda.current_stream(device_index).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"lh_selftest_fp16_subnormal failed on cuda:{device_index} ({ERRORS.get(rc, rc)}): fp16 subnormal "
                           "MFMA operands are flushed; the split-precision kernels would lose precision silently")
    _selftested.add(device_index)


def load() -> Lib:
    """The gfx950 library (cached). Raises if it has not been built."""
    global _hip_lib
    if _hip_lib is None:
        # LOOKONCE_HIP_LIB: another build of the same ABI (A/B timing of kernel revisions inside one gpurun call)
        _hip_lib = Lib(os.environ.get("LOOKONCE_HIP_LIB", HIP_LIB_PATH))
        # LOOKONCE_TUNE="key=value,...": lh_set_tuning switches applied at load (A/B runs of the whole test suite)
        for kv in filter(None, os.environ.get("LOOKONCE_TUNE", "").split(",")):
            k, v = kv.split("=")
            _hip_lib.call("lh_set_tuning", int(k), int(v))
    return _hip_lib
