"""CPU: checkpoint import/export adjacent to the path — Lightning-style state dicts load
strict into the drop-ins, and the packed weight blob round-trips bit-exactly with the documented header."""
import struct
import subprocess
import sys

import torch

from lookoncetohear_amd import _cabi, checkpoint
from lookoncetohear_amd.embed_net import EmbedTFGridNet
from lookoncetohear_amd.net import Net
from oracle import embedder_oracle as E
from oracle import tfgridnet_oracle as O


def test_lightning_checkpoint_roundtrip(tmp_path, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    # what the reference's trainer writes: PLModule.state_dict() = {"model.<net key>": tensor} (+ unrelated entries)
    ckpt = {"state_dict": {**{"model." + k: v for k, v in sd.items()}, "loss_fn.dummy": torch.zeros(1)}, "epoch": 3}
    path = tmp_path / "best.ckpt"
    torch.save(ckpt, path)
    net = checkpoint.load_lightning_checkpoint(str(path), Net(**O.TSH_PARAMS))
    assert not net.training
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k
    back = checkpoint.to_lightning_state_dict(net)
    assert set(back) == {"model." + k for k in sd}
    # embedder: same mechanism, `model.` prefix of the embedding PL module
    ecfg = E.ECfg(**E.EMBED_PARAMS)
    esd = E.synthetic_state_dict(ecfg, 1)
    enet = checkpoint.load_lightning_checkpoint({"state_dict": {"model." + k: v for k, v in esd.items()}},
                                                EmbedTFGridNet(**E.EMBED_PARAMS))
    assert all(torch.equal(v, esd[k]) for k, v in enet.state_dict().items())


def test_packed_blob_roundtrip(tmp_path, oracle_cfg_sd):
    cfg, sd = oracle_cfg_sd
    net = Net(**O.TSH_PARAMS).eval()
    net.load_state_dict(sd, strict=True)
    path = str(tmp_path / "tsh.lhw")
    meta = checkpoint.export_packed(net, path)
    raw = open(path, "rb").read()
    assert raw[:8] == b"LHWPACK1"
    abi, n = struct.unpack("<II", raw[8:16])
    assert abi == _cabi.ABI_VERSION and meta["model"] == "separator"
    assert {k: meta["params"][k] for k in O.TSH_PARAMS} == O.TSH_PARAMS and meta["params"]["num_src"] == 2
    start = (16 + n + 255) // 256 * 256
    assert len(raw) == start + meta["payload_bytes"]
    assert all(e["offset"] % 256 == 0 for e in meta["tensors"])
    kind, params, tensors = checkpoint.packed_tensors(net)
    meta2, loaded = checkpoint.import_packed(path)
    assert set(loaded) == set(tensors) and meta2["abi_version"] == abi
    for k, t in tensors.items():
        assert loaded[k].dtype == t.dtype and torch.equal(loaded[k], t), k
    # the names are the C-ABI argument images, e.g. the fused intra LSTM weights and the analysis filterbank
    assert "blocks.0.intra_w16" in loaded and "wfb_t" in loaded
    out = subprocess.run([sys.executable, "-m", "lookoncetohear_amd.checkpoint", "info", path], capture_output=True, text=True)
    assert out.returncode == 0 and "separator ABI v%d" % abi in out.stdout

    enet = EmbedTFGridNet(**E.EMBED_PARAMS).eval()
    epath = str(tmp_path / "embed.lhw")
    emeta = checkpoint.export_packed(enet, epath)
    _, eloaded = checkpoint.import_packed(epath)
    _, _, etensors = checkpoint.packed_tensors(enet)
    assert emeta["model"] == "embedder" and all(torch.equal(eloaded[k], v) for k, v in etensors.items())
