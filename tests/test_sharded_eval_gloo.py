"""CPU, world_size 2, gloo: the N>1 utterance-sharded eval path (shard -> per-rank metric sums -> one all-reduce)
gives the same aggregate as the single-process run — through the enrollment branch of the loop (embedding from
`inputs['enrollments']`, reference src/ts_hear_test.py:132-135; a stand-in embedder here).  The separator is replaced by the CPU oracle on short clips
(the HIP path cannot run here); what is under test is the sharding / reduction plumbing of lookoncetohear_amd.eval."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from lookoncetohear_amd import synth
from lookoncetohear_amd.eval import evaluate
from oracle import tfgridnet_oracle as O
torch.set_num_threads(2)
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = O.Cfg(**O.TSH_PARAMS); sd = O.synthetic_state_dict(cfg, 0)
model = lambda m, e: O.forward(cfg, sd, m, e)
# enrollment branch of the loop (reference src/ts_hear_test.py:132-135) with a stand-in embedder: unit-norm mean spectrum
enroll = lambda x: torch.nn.functional.normalize(x.reshape(x.shape[0], -1)[:, :256].abs() + 1e-3, dim=-1)
agg, rows = evaluate(model, lambda idx: synth.batch(idx, 1500, enroll_n=600), n_utts=5, batch_size=2, rank=rank, world=world,
                     dist=dist, enroll_model=enroll, all_rows=True)
if rank == 0:
    print("RESULT " + json.dumps(agg))
print("ROWS%%d " %% rank + json.dumps(rows))              # the gathered table, on every rank
dist.destroy_process_group()
""" % ROOT


def test_world2_matches_world1(tmp_path):
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.eval import evaluate, shard_indices
    from oracle import tfgridnet_oracle as O
    assert sorted(shard_indices(5, 0, 2) + shard_indices(5, 1, 2)) == list(range(5))
    cfg = O.Cfg(**O.TSH_PARAMS)
    sd = O.synthetic_state_dict(cfg, 0)
    enroll = lambda x: torch.nn.functional.normalize(x.reshape(x.shape[0], -1)[:, :256].abs() + 1e-3, dim=-1)
    ref, rows = evaluate(lambda m, e: O.forward(cfg, sd, m, e), lambda idx: synth.batch(idx, 1500, enroll_n=600), n_utts=5,
                         batch_size=2, enroll_model=enroll)
    assert ref["n"] == 5 and len(rows) == 5
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    import re                      # gloo writes its own "[Gloo] Rank ..." lines to stdout, unbuffered: do not rely on line starts
    got = json.loads(re.search(r"RESULT (\{.*?\})", out.stdout).group(1))
    assert got["n"] == 5
    for k in ("si_snr_i", "output_sisnr", "embedding_sim"):
        assert abs(got[k] - ref[k]) < 1e-4, (k, got[k], ref[k])
    # the per-utterance table across ranks (reference CSV, src/ts_hear_test.py:149-151, 162-166): every rank holds all 5 rows
    # in utterance order, equal to the single-process rows
    for r in (0, 1):
        table = json.loads(re.search(r"ROWS%d (\[.*?\])" % r, out.stdout).group(1))
        assert [t["idx"] for t in table] == [0, 1, 2, 3, 4]
        for t, q in zip(table, sorted(rows, key=lambda q: q["idx"])):
            for k in ("output_sisnr", "si_snr_i", "embedding_sim"):
                assert abs(t[k] - q[k]) < 1e-4, (r, k, t, q)


RANGE_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from lookoncetohear_amd import synth
from lookoncetohear_amd.eval import evaluate
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
calls = [0]
def model(m, e):
    # a Net with the default deferred range check raises from the forward AFTER the bad batch (net.py `_separate`): here
    # rank 0's second of three batches, i.e. NOT its last one
    calls[0] += 1
    if rank == 0 and calls[0] == 2:
        raise RuntimeError("LH_ERR_RANGE: an earlier forward of this Net produced non-finite samples.")
    return 0.6 * m
try:
    evaluate(model, lambda idx: synth.batch(idx, 1500), n_utts=12, batch_size=2, rank=rank, world=world, dist=dist)
    print("NORAISE%%d" %% rank)
except RuntimeError as e:
    print("RAISED%%d %%s" %% (rank, str(e)[:40]))
dist.destroy_process_group()
""" % ROOT


def test_range_error_mid_loop_reaches_every_rank_after_the_collective(tmp_path):
    """a deferred LH_ERR_RANGE raised by a forward in the MIDDLE of rank 0's shard used to leave the loop
    before the all-reduce, and rank 1 hung in the collective.  Now rank 0 stops, poisons its sums, enters the all-reduce and
    both ranks raise after it."""
    script = tmp_path / "worker_range.py"
    script.write_text(RANGE_WORKER)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RAISED0 LH_ERR_RANGE" in out.stdout and "RAISED1 LH_ERR_RANGE" in out.stdout, out.stdout


def test_metrics_match_oracle_definition():
    from lookoncetohear_amd.metrics import si_snr
    from oracle import tfgridnet_oracle as O
    g = torch.Generator().manual_seed(1)
    t = torch.randn(3, 2, 3000, generator=g)
    p = 0.5 * t + 0.2 * torch.randn(3, 2, 3000, generator=g)
    assert torch.allclose(si_snr(p, t), O.si_snr(p, t), atol=1e-5)
