"""Utterance-sharded evaluation, shaped like the reference eval driver (reference src/ts_hear_test.py:93-166)
with its hot loop (:124-150) kept: for each batch  embedding -> model(mixture, embedding) -> SI-SNR / SI-SNRi /
cosine similarity per utterance.  The reference is single-process; here utterances are independent (eval mode,
per-sample norms, per-utterance state), so they are partitioned `idx % world == rank` over one process per GPU
and the only exchange step is ONE all-reduce of `[sum si_snr_i, sum output_sisnr, sum embedding_sim, n]`
(RCCL over xGMI on the GPUs, gloo in the CPU tests) — SURVEY.md §8(e) — plus, on request (`all_rows`), the all-gather of
the per-utterance rows that rebuilds the reference's CSV table (src/ts_hear_test.py:162-166).  No data-path collective.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional

import torch

from .metrics import metric_sums, metric_sums_device, per_utterance


def shard_indices(n_utts: int, rank: int, world: int) -> List[int]:
    """Strided split; inputs are seeded by utterance index so the union is independent of `world`."""
    return list(range(rank, n_utts, world))


def gather_rows(rows: List[dict], n_utts: int, world: int, device, dist) -> List[dict]:
    """The per-utterance table of the reference CSV (src/ts_hear_test.py:149-151, 162-166) across ranks: an all-gather of
    `(idx, output_sisnr, si_snr_i, embedding_sim)` — 4 x fp64 x ceil(n / world) per rank, SURVEY.md §8(e)'s optional second
    exchange — returned in utterance order on every rank.  Ranks own ceil or floor(n / world) utterances: shorter shards pad
    with idx = -1 rows, dropped after the collective."""
    cap = (n_utts + world - 1) // world
    mine = torch.full((cap, 4), -1.0, dtype=torch.float64, device=device)
    if rows:
        mine[:len(rows)] = torch.tensor([[r["idx"], r["output_sisnr"], r["si_snr_i"], r["embedding_sim"]] for r in rows],
                                        dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    table = torch.cat(parts).cpu()
    table = table[table[:, 0] >= 0]
    table = table[torch.argsort(table[:, 0])]
    return [dict(idx=int(r[0]), output_sisnr=float(r[1]), si_snr_i=float(r[2]), embedding_sim=float(r[3])) for r in table.tolist()]


def evaluate(model: Callable, data_fn: Callable[[List[int]], dict], n_utts: int, batch_size: int = 4,
             rank: int = 0, world: int = 1, device="cpu", dist=None, enroll_model: Optional[Callable] = None,
             range_status: Optional[Callable[[], bool]] = None, all_rows: bool = False):
    """Returns (mean si_snr_i, mean output_sisnr, mean embedding_sim, n) over ALL ranks, plus this rank's rows — or, with
    `all_rows=True`, the rows of ALL ranks in utterance order (`gather_rows`: the table the reference writes to its CSV).

    `model(mixture [B,2,N], embedding [B,1,256]) -> [B,2,N]` is the separator (`Net.forward`);
    `data_fn(indices)` returns the dict of reference dataset fields (mixture, target, embedding_gt[, enrollments]);
    `range_status`: the separator's `Net.range_status` when `model` is a wrapper around it (a closure has none; a `Net` or its
    bound `forward` is found by itself).  Non-finite data cannot go unnoticed: the offline forward hands inf / NaN through
    like the reference (`keep_nonfinite`), so the metric sums of such a batch are NaN, and a `Net` with the default deferred
    range check raises LH_ERR_RANGE from its NEXT forward — that raise is caught here (it used to leave
    the loop BEFORE the collective and the other ranks waiting in it), the rank stops its shard, poisons its sums and still
    enters the all-reduce; every rank then raises AFTER the collective.
    """
    mine = shard_indices(n_utts, rank, world)
    total = torch.zeros(4, dtype=torch.float64, device=device)
    rows = []
    bad = False
    with torch.no_grad():
        for s in range(0, len(mine), batch_size):
            idx = mine[s:s + batch_size]
            d = data_fn(idx)
            mixture = d["mixture"].to(device)
            target = d["target"].to(device)
            emb_gt = d["embedding_gt"].to(device)
            try:
                if enroll_model is not None:                     # ts_hear_test.py:132-135
                    enrollments = d["enrollments"]
                    if enrollments.dim() == 4:                   # [B, num_enroll = 1, 2, N] as the dataset returns it
                        enrollments = enrollments.squeeze(1)
                    embedding = enroll_model(enrollments.to(device)).unsqueeze(1)
                else:
                    embedding = emb_gt                           # :137
                outputs = model(mixture, embedding)              # :138  <- the hot path
            except RuntimeError as e:                            # the deferred look at an EARLIER batch's flag (net.py)
                if "LH_ERR_RANGE" not in str(e):
                    raise
                bad = True
                break
            if outputs.is_cuda:                              # HIP metric kernels: no waveform leaves the device
                sums, r = metric_sums_device(outputs, mixture, target, embedding[:, 0], emb_gt[:, 0])
                total += sums
                o, i, c = r[:, 0], r[:, 1], r[:, 2]
            else:                                            # host tensors (CPU tests of the sharding plumbing)
                total += metric_sums(outputs, mixture, target, embedding[:, 0], emb_gt[:, 0])
                o, i, c = per_utterance(outputs, mixture, target, embedding[:, 0], emb_gt[:, 0])
            rows += [dict(idx=k, output_sisnr=float(a), si_snr_i=float(b), embedding_sim=float(e))
                     for k, a, b, e in zip(idx, o.tolist(), i.tolist(), c.tolist())]
    # range guard (include/lookonce_hip.h): a Net looks at its flag when the NEXT forward starts; after the last batch ask it
    if range_status is None:
        net = getattr(model, "__self__", model)              # a bound method (net.forward) or the module itself
        range_status = getattr(net, "range_status", None)
    bad = (bool(range_status()) if callable(range_status) else False) or bad
    if bad:
        total[:3] = float("nan")                             # the other ranks learn it from the all-reduced sums
    if dist is not None and world > 1:
        dist.all_reduce(total)                               # sum over ranks, 32 bytes
    if bad or not bool(torch.isfinite(total).all()):
        raise RuntimeError("LH_ERR_RANGE: a batch of this evaluation produced non-finite samples (inf / NaN in the input)"
                           + ("" if bad else " on another rank, or its metrics are not finite"))
    if all_rows and dist is not None and world > 1:          # every rank is past the raise above or none is
        rows = gather_rows(rows, n_utts, world, device, dist)
    n = max(float(total[3].item()), 1.0)
    return dict(si_snr_i=float(total[0].item()) / n, output_sisnr=float(total[1].item()) / n,
                embedding_sim=float(total[2].item()) / n, n=int(total[3].item())), rows
