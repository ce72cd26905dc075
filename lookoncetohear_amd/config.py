"""Model configurations of the reference (`configs/tsh.json:5-19`, `configs/embed.json:5-11`) and deterministic
random-init weights of those architectures.

No checkpoint exists in the reference tree (`*.ckpt` is git-ignored) and there is no network, so `bench.py`,
`__graft_entry__.smoke()` and the tests run on random-init weights of the configured architecture.  The generator is
name-keyed (seed derived from the position of the tensor name in the sorted state-dict) so the same weights come out on
any box; scales follow torch's default initialisers, with norm affines / PReLU slopes / biases perturbed away from their
1 / 0 / 0.25 defaults so that every learned tensor influences the output.  `oracle/` carries its own copy of the same
rules (it must stay self-contained); `tests/test_oracle_golden.py` asserts the two agree bit for bit.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

# configs/tsh.json "model_params" (the separator) and configs/embed.json "model_params" (the enrollment embedder)
TSH_PARAMS = dict(embed_dim=256, stft_chunk_size=128, stft_pad_size=64, num_ch=2, D=64, L=4, I=1, J=1,
                  B=3, H=64, local_atten_len=50, use_attn=True, lookahead=True, chunk_causal=True)
EMBED_PARAMS = dict(embed_dim=256, num_ch=2, n_fft=128, stride=64, num_blocks=3)


def _uniform(shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1


def separator_weights(seed: int = 0, params: dict = TSH_PARAMS) -> Dict[str, torch.Tensor]:
    """Random-init state dict of `lookoncetohear_amd.net.Net(**params)` (= the reference `Net`'s names and shapes)."""
    from .net import Net, stft_filterbank
    net = Net(**params)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = {}
    for idx, name in enumerate(sorted(shapes)):
        shape = shapes[name]
        if name.endswith("_filters"):
            sd[name] = stft_filterbank(net.nfft, net.stft_chunk_size)
            continue
        u = _uniform(shape, seed * 100003 + idx)
        if ".norm.weight" in name or name.endswith("embed_to_feats_proj.1.weight"):
            t = 1.0 + 0.25 * u
        elif ".norm.bias" in name or name.endswith("embed_to_feats_proj.1.bias"):
            t = 0.1 * u
        elif name.endswith(".1.weight") and shape == (1,):
            t = 0.25 + 0.1 * u                      # PReLU slope
        else:
            if len(shape) == 1:                     # biases
                fan_in = 64
            elif "deconv.weight" in name:           # ConvTranspose2d sums over in_ch * k * k terms
                fan_in = shape[0] * shape[2] * shape[3]
            else:
                fan_in = int(np.prod(shape[1:]))
            t = u / math.sqrt(fan_in)
        sd[name] = t.float()
    return sd


def embedder_weights(seed: int = 0, params: dict = EMBED_PARAMS) -> Dict[str, torch.Tensor]:
    """Random-init state dict of `lookoncetohear_amd.embed_net.EmbedTFGridNet(**params)`."""
    from .embed_net import EmbedTFGridNet
    shapes = {k: tuple(v.shape) for k, v in EmbedTFGridNet(**params).state_dict().items()}
    sd = {}
    for idx, name in enumerate(sorted(shapes)):
        shape = shapes[name]
        u = _uniform(shape, seed * 100003 + 7919 + idx)
        if name.endswith(".gamma") or name in ("conv.1.weight", "embed_proj.1.weight"):
            t = 1.0 + 0.25 * u
        elif name.endswith(".beta") or name in ("conv.1.bias", "embed_proj.1.bias"):
            t = 0.1 * u
        elif name.endswith(".1.weight") and shape == (1,):
            t = 0.25 + 0.1 * u
        elif len(shape) == 1:
            t = u / 8.0
        else:
            fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1) * (shape[3] if len(shape) > 3 else 1)
            if name.endswith("_linear.weight"):     # ConvTranspose1d [in, out, k]
                fan_in = shape[0] * shape[2]
            t = u / math.sqrt(fan_in)
        sd[name] = t.float()
    return sd
