"""TEST TOOLING: the product host classes driven over the emulated library with host tensors.

`lookoncetohear_amd._cabi.HipHost` concentrates the device plumbing of `Net`, `EmbedTFGridNet` and `BinauralRenderer` in
four methods (library, launch stream, current-device context, flag memory) plus `Net`'s two stream queries; the
subclasses below override exactly those, so every other line of the host code that the CPU tests execute is the line
the GPU runs.  The product classes themselves carry no test hook and refuse host tensors."""
import contextlib

import torch

from lookoncetohear_amd import _cabi
from lookoncetohear_amd.embed_net import EmbedTFGridNet
from lookoncetohear_amd.net import Net
from lookoncetohear_amd.render import BinauralRenderer


class EmuHost(_cabi.HipHost):
    emu_lib = None                          # a `_cabi.Lib` over tests/hipemu/_build/liblookonce_emu.so

    def _lib(self, t):
        assert self.emu_lib is not None and not t.is_cuda
        return self.emu_lib

    def _stream(self, device):
        return 0

    def _device_ctx(self, t):
        return contextlib.nullcontext()

    def _flag_words(self, device):
        return torch.zeros(2, dtype=torch.int32)


class SerialLanes:
    """Stand-in for `net._Lanes` (the K streams of `Net.time_chunks`): the emulator executes every launch at once, in
    enqueue order — block-major, window-minor, which satisfies every dependence the events express."""
    serial = True

    def fork(self): pass
    def on(self, k): return contextlib.nullcontext()
    def signal(self, k): return None
    def wait(self, k, ev): pass
    def join(self): pass


class EmuNet(EmuHost, Net):
    def _sync(self, dev):
        pass

    def _lanes(self, dev, K):
        return SerialLanes()

    def _capturing(self):
        return False


class EmuEmbed(EmuHost, EmbedTFGridNet):
    def _multi_stream(self, input):
        return False


class EmuRenderer(EmuHost, BinauralRenderer):
    pass


class EmuMetricHost(EmuHost):
    def __init__(self, lib):
        self.emu_lib = lib
