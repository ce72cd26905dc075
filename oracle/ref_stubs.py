"""TEST INFRASTRUCTURE — not part of the product path.

Stub modules that let the *unmodified* reference separator
(`/root/reference/src/models/tfgridnet_realtime/net.py`) be imported in this
container, where `espnet2` and `asteroid_filterbanks` are not installed
(SURVEY.md §8c, Appendix B).  Only `oracle/gen_golden.py` and
`oracle/check_against_reference.py` use this, and only here: `/root/reference`
does not exist on the GPU box, so nothing at test/bench time imports this file.

Six names are needed by `tfgridnet_causal.py:12-18`; five are trivial, the
sixth (`asteroid_filterbanks.make_enc_dec`) carries real arithmetic and is
restated from the published asteroid-filterbanks STFTFB definition
(un-pinned third-party dependency: `requirements.txt:15` lists `asteroid`
with no version; restated algorithm = asteroid-filterbanks 0.4.0 `STFTFB` +
`Encoder`/`Decoder`).
"""
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def stft_filterbank(n_filters: int, kernel_size: int, stride: int) -> torch.Tensor:
    """asteroid `STFTFB` filters, shape [n_filters + 2, 1, kernel_size].

    window = sqrt(periodic hann); filters = rfft basis rows (real then imag),
    DC and Nyquist real rows additionally divided by sqrt(2); overall scale
    1 / (0.5 * sqrt(kernel_size * n_filters / stride)).
    The `window_type` kwarg the reference passes (`tfgridnet_causal.py:135`)
    is swallowed by asteroid's **kwargs, so the default window is used.
    """
    assert n_filters == kernel_size and n_filters % 2 == 0
    cutoff = n_filters // 2 + 1
    window = np.hanning(kernel_size + 1)[:-1] ** 0.5
    lpad = int((n_filters - kernel_size) // 2)
    rpad = int(n_filters - kernel_size - lpad)
    scale = 0.5 * np.sqrt(kernel_size * n_filters / stride)
    basis = np.fft.fft(np.eye(n_filters))
    basis = basis[:, lpad:(n_filters - rpad)] if rpad or lpad else basis
    filters = np.vstack([np.real(basis[:cutoff, :]), np.imag(basis[:cutoff, :])])
    filters[0, :] /= np.sqrt(2)
    filters[n_filters // 2, :] /= np.sqrt(2)
    filters = filters / scale * window
    return torch.from_numpy(filters).unsqueeze(1).float()


class _FB(nn.Module):
    def __init__(self, n_filters, kernel_size, stride):
        super().__init__()
        self.n_filters, self.kernel_size, self.stride = n_filters, kernel_size, stride
        self.register_buffer("_filters", stft_filterbank(n_filters, kernel_size, stride))

    def filters(self):
        return self._filters


class _Encoder(nn.Module):
    """asteroid `Encoder`: conv1d(x.view(-1,1,N), filters, stride), batch dims kept."""

    def __init__(self, fb):
        super().__init__()
        self.filterbank = fb
        self.stride = fb.stride

    def forward(self, x):
        shp = x.shape
        y = F.conv1d(x.reshape(-1, 1, shp[-1]), self.filterbank.filters(), stride=self.stride)
        return y.view(*shp[:-1], y.shape[-2], y.shape[-1])


class _Decoder(nn.Module):
    """asteroid `Decoder`: conv_transpose1d(spec.view(-1,K,T), filters, stride)."""

    def __init__(self, fb):
        super().__init__()
        self.filterbank = fb
        self.stride = fb.stride

    def forward(self, spec):
        shp = spec.shape
        y = F.conv_transpose1d(spec.reshape(-1, shp[-2], shp[-1]), self.filterbank.filters(),
                               stride=self.stride)
        return y.view(*shp[:-2], -1)


def make_enc_dec(fb_name, n_filters, kernel_size, stride=None, **kwargs):
    assert fb_name == "stft"
    return (_Encoder(_FB(n_filters, kernel_size, stride)),
            _Decoder(_FB(n_filters, kernel_size, stride)))


def _get_layer(name):
    for k in dir(nn):
        if k.lower() == name.lower():
            return getattr(nn, k)
    raise KeyError(name)


def install():
    """Inject the stub modules and put the reference root on sys.path."""
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

    def mod(path, **attrs):
        parts = path.split(".")
        for i in range(1, len(parts) + 1):
            p = ".".join(parts[:i])
            if p not in sys.modules:
                m = types.ModuleType(p)
                m.__path__ = []
                sys.modules[p] = m
        for k, v in attrs.items():
            setattr(sys.modules[path], k, v)

    class _Placeholder:
        pass

    class AbsSeparator(nn.Module):
        pass

    mod("espnet2.enh.decoder.stft_decoder", STFTDecoder=_Placeholder)
    mod("espnet2.enh.encoder.stft_encoder", STFTEncoder=_Placeholder)
    mod("espnet2.enh.layers.complex_utils", new_complex_like=_Placeholder)
    mod("espnet2.enh.separator.abs_separator", AbsSeparator=AbsSeparator)
    mod("espnet2.torch_utils.get_layer_from_string", get_layer=_get_layer)
    mod("asteroid_filterbanks", make_enc_dec=make_enc_dec)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_net_class():
    install()
    from src.models.tfgridnet_realtime.net import Net  # noqa: the unmodified reference
    return Net
