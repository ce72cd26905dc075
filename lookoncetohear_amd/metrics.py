"""Eval metrics of the reference test loop (reference src/ts_hear_test.py:140-146), restated on torch tensors
so they run on the device that holds the separator output (no `outputs.cpu()` round trip).

`scale_invariant_signal_noise_ratio` (torchmetrics, absent here) = zero-mean SI-SDR:
    alpha = (<p,t> + eps) / (<t,t> + eps);  10 log10((|alpha t|^2 + eps) / (|alpha t - p|^2 + eps)), eps = fp32 eps.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def si_snr(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    eps = torch.finfo(pred.dtype).eps
    pred = pred - pred.mean(-1, keepdim=True)
    target = target - target.mean(-1, keepdim=True)
    alpha = ((pred * target).sum(-1, keepdim=True) + eps) / ((target * target).sum(-1, keepdim=True) + eps)
    scaled = alpha * target
    noise = scaled - pred
    return 10.0 * torch.log10(((scaled * scaled).sum(-1) + eps) / ((noise * noise).sum(-1) + eps))


def per_utterance(outputs, mixture, target, embedding, embedding_gt):
    """Rows of the reference CSV (ts_hear_test.py:149-151): output_sisnr, si_snr_i, embedding_sim — each [B]."""
    out_sisnr = si_snr(outputs, target)                                   # [B, 2]
    snr_i = out_sisnr - si_snr(mixture, target)
    return (out_sisnr.mean(dim=1), snr_i.reshape(snr_i.shape[0], -1).mean(dim=1),
            F.cosine_similarity(embedding, embedding_gt, dim=-1))


def metric_sums(outputs, mixture, target, embedding, embedding_gt) -> torch.Tensor:
    """[sum si_snr_i, sum output_sisnr, sum embedding_sim, n] as fp64 on the outputs' device — the 32-byte
    payload of the sharded eval's single all-reduce."""
    o, i, c = per_utterance(outputs, mixture, target, embedding, embedding_gt)
    n = torch.tensor(float(outputs.shape[0]), device=outputs.device, dtype=torch.float64)
    return torch.stack([i.double().sum(), o.double().sum(), c.double().sum(), n])


def metric_sums_device(outputs, mixture, target, embedding, embedding_gt, host=None):
    """Same quantities through the HIP kernels of lh_metrics.hip (fp64 moments, one pass over the waveforms).
    Returns (sums [4] fp64 on device, rows [B,3] fp32 = output_sisnr, si_snr_i, embedding_sim).  `host`: the
    `_cabi.HipHost` whose library / stream plumbing to use (default: the product library on the tensors' GPU)."""
    from . import _cabi
    if host is None:
        host = type("MetricHost", (_cabi.HipHost,), {"_host_name": "metric_sums_device"})()
    lib = host._lib(outputs)
    B, _, n = outputs.shape
    dev = outputs.device
    c32 = lambda t: t.contiguous().float()
    o, m, t = c32(outputs), c32(mixture), c32(target)
    e, g = c32(embedding.reshape(B, -1)), c32(embedding_gt.reshape(B, -1))
    scratch = torch.empty(B * 2 * 16 * 8 + B * 3, dtype=torch.float64, device=dev)
    rows = torch.empty(B, 3, dtype=torch.float32, device=dev)
    sums = torch.empty(4, dtype=torch.float64, device=dev)
    st = host._stream(dev)
    with host._device_ctx(o):
        lib.call("lh_metric_sums", o.data_ptr(), t.data_ptr(), m.data_ptr(), e.data_ptr(), g.data_ptr(),
                 scratch.data_ptr(), rows.data_ptr(), sums.data_ptr(), B, n, e.shape[1], st)
    return sums, rows
