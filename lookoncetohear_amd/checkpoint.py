"""Checkpoint import / export adjacent to the hot path.

* `load_lightning_checkpoint` — what the reference's `load_model` does (`src/ts_hear_test.py:18-34`:
  `torch.load(run_dir/best.ckpt)['state_dict']` into the Lightning module, whose separator / embedder lives under the
  attribute `model`, `src/ts_hear_embed_pl_module.py:25`): strips the `model.` prefix and loads `strict=True` into the
  drop-in `Net` / `EmbedTFGridNet`, so a reference-trained checkpoint runs on the HIP path unchanged.
* `export_packed` / `import_packed` — the packed weight blob for hosts that call the C ABI without Python: one flat
  little-endian file holding every tensor the kernels consume, already in their MFMA fragment order (the output of
  `weights.pack_all` / `embed_net.pack_embedder`), each 256-byte aligned so the whole payload can be copied to the
  device in one `hipMemcpy` and addressed by offset.  Format (`include/lookonce_weights.h`):

      bytes 0..7    magic  "LHWPACK1"
      bytes 8..11   uint32 ABI version (lh_abi_version() the images were packed for)
      bytes 12..15  uint32 length L of the JSON index
      bytes 16..16+L  JSON  {"model": "separator"|"embedder", "params": {...ctor kwargs...},
                             "tensors": [{"name": "blocks.0.intra_w16", "dtype": "float16", "shape": [...],
                                          "offset": <bytes from payload start>, "nbytes": ...}, ...]}
      padding to a multiple of 256, then the payload

      python -m lookoncetohear_amd.checkpoint export runs/tsh/best.ckpt tsh.lhw --config configs/tsh.json
      python -m lookoncetohear_amd.checkpoint info tsh.lhw
"""
from __future__ import annotations

import json
import struct
import sys
from typing import Dict, Tuple

import numpy as np
import torch

MAGIC = b"LHWPACK1"
ALIGN = 256
_DTYPES = {"float32": (torch.float32, np.float32), "float16": (torch.float16, np.float16),
           "float64": (torch.float64, np.float64), "int32": (torch.int32, np.int32)}


def strip_prefix(state_dict: Dict[str, torch.Tensor], prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """Lightning-module keys -> model keys (`model.tfgridnet.conv.0.weight` -> `tfgridnet.conv.0.weight`); keys without
    the prefix (loss modules, metrics) are dropped, exactly the sub-tree `PLModule.model` owns."""
    out = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    if not out:
        raise KeyError(f"no key starts with {prefix!r}; first keys: {list(state_dict)[:3]}")
    return out


def load_lightning_checkpoint(ckpt, module: torch.nn.Module, prefix: str = "model.", strict: bool = True):
    """ckpt: path or an already loaded dict with a 'state_dict' entry.  Returns `module` (eval mode)."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    if any(k.startswith(prefix) for k in sd):
        sd = strip_prefix(sd, prefix)
    module.load_state_dict(sd, strict=strict)
    return module.eval()


def to_lightning_state_dict(module: torch.nn.Module, prefix: str = "model.") -> Dict[str, torch.Tensor]:
    """The inverse: a state dict the reference's `PLModule.load_state_dict` accepts."""
    return {prefix + k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _flatten(tree, pre="") -> Dict[str, torch.Tensor]:
    out = {}
    if isinstance(tree, dict):
        for k, v in tree.items():
            out.update(_flatten(v, f"{pre}{k}."))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            out.update(_flatten(v, f"{pre}{i}."))
    else:
        out[pre[:-1]] = tree
    return out


def packed_tensors(module: torch.nn.Module) -> Tuple[str, dict, Dict[str, torch.Tensor]]:
    """(model kind, ctor kwargs, flat name -> packed tensor) for a `Net` or an `EmbedTFGridNet`."""
    from .embed_net import EmbedTFGridNet, pack_embedder
    from .net import Net
    from .weights import pack_all
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    if isinstance(module, Net):
        return "separator", dict(module.ctor_params), _flatten(pack_all(sd, module.n_blocks))
    if isinstance(module, EmbedTFGridNet):
        params = dict(embed_dim=module.embed_dim, num_ch=module.n_imics, n_fft=module.n_fft, stride=module.stride,
                      num_blocks=module.n_layers)
        return "embedder", params, _flatten(pack_embedder(sd, module.n_layers))
    raise TypeError(type(module))


def export_packed(module: torch.nn.Module, path: str) -> dict:
    from . import _cabi
    kind, params, tensors = packed_tensors(module)
    index, off = [], 0
    for name, t in tensors.items():
        t = t.contiguous()
        dt = str(t.dtype).replace("torch.", "")
        if dt not in _DTYPES:
            raise TypeError(f"{name}: {t.dtype}")
        nbytes = t.numel() * t.element_size()
        index.append({"name": name, "dtype": dt, "shape": list(t.shape), "offset": off, "nbytes": nbytes})
        off = (off + nbytes + ALIGN - 1) // ALIGN * ALIGN
    meta = {"model": kind, "params": params, "tensors": index, "payload_bytes": off}
    js = json.dumps(meta).encode()
    head = MAGIC + struct.pack("<II", _cabi.ABI_VERSION, len(js)) + js
    head += b"\0" * ((-len(head)) % ALIGN)
    with open(path, "wb") as f:
        f.write(head)
        pos = 0
        for e, t in zip(index, tensors.values()):
            f.write(b"\0" * (e["offset"] - pos))
            f.write(t.contiguous().numpy().tobytes())
            pos = e["offset"] + e["nbytes"]
        f.write(b"\0" * (off - pos))
    return meta


def read_index(path: str) -> Tuple[dict, int]:
    with open(path, "rb") as f:
        head = f.read(16)
        if head[:8] != MAGIC:
            raise ValueError(f"{path}: not a LHWPACK1 file")
        abi, n = struct.unpack("<II", head[8:16])
        meta = json.loads(f.read(n))
    meta["abi_version"] = abi
    start = (16 + n + ALIGN - 1) // ALIGN * ALIGN
    return meta, start


def import_packed(path: str, device="cpu") -> Tuple[dict, Dict[str, torch.Tensor]]:
    """Reads the blob back: (index, flat name -> tensor).  With a GPU `device` the payload is uploaded in one copy
    and the returned tensors are views into that single allocation (the layout a C host would use)."""
    meta, start = read_index(path)
    raw = np.fromfile(path, dtype=np.uint8, offset=start, count=meta["payload_bytes"])
    buf = torch.from_numpy(raw).to(device)
    out = {}
    for e in meta["tensors"]:
        tdt, _ = _DTYPES[e["dtype"]]
        out[e["name"]] = buf[e["offset"]:e["offset"] + e["nbytes"]].view(tdt).view(*e["shape"])
    return meta, out


def run_packed(path: str, x: torch.Tensor, embed: torch.Tensor) -> torch.Tensor:
    """Offline separation `x [B, 2, N]`, `embed [B, 256]` -> `[B, 2, N]` driven by a packed blob alone (no state dict,
    no parameter tree): what a C host does with `include/lookonce_weights.h` + `include/lookonce_hip.h`."""
    from .net import Net
    net = Net.from_packed(path, x.device)
    with torch.no_grad():
        y, _ = net.predict(x, embed, None, pad=True, want_state=False)
    return y


def _main(argv):
    if len(argv) >= 2 and argv[0] == "info":
        meta, start = read_index(argv[1])
        print(f"{argv[1]}: {meta['model']} ABI v{meta['abi_version']} params {meta['params']}")
        print(f"payload at byte {start}, {meta['payload_bytes']} bytes, {len(meta['tensors'])} tensors")
        for e in meta["tensors"]:
            print(f"  {e['offset']:>10d} {e['nbytes']:>9d} {e['dtype']:8s} {e['shape']} {e['name']}")
        return 0
    if len(argv) >= 3 and argv[0] == "export":
        cfg_path = argv[argv.index("--config") + 1] if "--config" in argv else None
        if cfg_path is None:
            raise SystemExit("export needs --config <configs/tsh.json | configs/embed.json>")
        cfg = json.load(open(cfg_path))["pl_module_args"]
        from .embed_net import EmbedTFGridNet
        from .net import Net
        cls = EmbedTFGridNet if cfg["model"].endswith("EmbedTFGridNet") else Net
        module = load_lightning_checkpoint(argv[1], cls(**cfg["model_params"]))
        meta = export_packed(module, argv[2])
        print(f"wrote {argv[2]}: {meta['model']}, {len(meta['tensors'])} tensors, {meta['payload_bytes']} bytes")
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(_main(sys.argv[1:]))
