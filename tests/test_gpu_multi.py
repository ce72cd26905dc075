"""GPU: the exchange step on real RCCL.

* single rank (runs on the 1-GPU box): `lh_comm_*` / `lh_allreduce_f64` of the C ABI create a one-rank RCCL communicator
  and all-reduce the metric sums in place on a stream (identity for one rank — what is exercised is the dlopen binding,
  the call signatures and the stream handling);
* two ranks (skipped unless 2 GPUs are visible): `bench.py --gpus 2` launched exactly as the driver launches the
  scaling runs (torch.distributed.run, backend nccl = RCCL), and the sharded eval of lookoncetohear_amd.eval on nccl
  against the single-process aggregate — both through the C-ABI all-reduce as well as torch.distributed's.
"""
import ctypes
import json
import os
import subprocess
import sys

import pytest
import torch

from lookoncetohear_amd import _cabi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_allreduce_single_rank():
    lib = _cabi.load()
    uid = (ctypes.c_char * 128)()
    rc = lib.raw("lh_comm_unique_id")(ctypes.cast(uid, ctypes.c_void_p))
    assert rc == 0, f"lh_comm_unique_id -> {rc} (2 = no RCCL found)"
    comm = ctypes.c_void_p()
    with torch.cuda.device(0):
        assert lib.raw("lh_comm_init")(ctypes.cast(uid, ctypes.c_void_p), 1, 0, ctypes.byref(comm)) == 0
        buf = torch.tensor([1.5, -2.25, 3.0, 4.0], dtype=torch.float64, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        assert lib.raw("lh_allreduce_f64")(comm, buf.data_ptr(), 4, st) == 0
        torch.cuda.synchronize()
        assert buf.tolist() == [1.5, -2.25, 3.0, 4.0]
        assert lib.raw("lh_comm_destroy")(comm) == 0


WORKER = r"""
import ctypes, os, sys, json, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from lookoncetohear_amd import _cabi, config, synth
from lookoncetohear_amd.eval import evaluate, gather_rows
from lookoncetohear_amd.net import Net
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group(backend="nccl", device_id=dev)
net = Net(**config.TSH_PARAMS).eval()
net.load_state_dict(config.separator_weights(0), strict=True)
net = net.to(dev)
agg, rows = evaluate(net, lambda idx: synth.batch(idx, 16000), n_utts=6, batch_size=2, rank=rank, world=world, device=dev, dist=dist)
# the same reduction through the C ABI (Python-free hosts): unique id broadcast over the existing group
lib = _cabi.load()
uid = torch.zeros(128, dtype=torch.uint8)
if rank == 0:
    raw = (ctypes.c_char * 128)()
    assert lib.raw("lh_comm_unique_id")(ctypes.cast(raw, ctypes.c_void_p)) == 0
    uid = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
uid = uid.to(dev); dist.broadcast(uid, 0); uid = uid.cpu().numpy().tobytes()
comm = ctypes.c_void_p()
assert lib.raw("lh_comm_init")(ctypes.c_char_p(uid), world, rank, ctypes.byref(comm)) == 0
mine = torch.tensor([sum(r["si_snr_i"] for r in rows), sum(r["output_sisnr"] for r in rows),
                     sum(r["embedding_sim"] for r in rows), float(len(rows))], dtype=torch.float64, device=dev)
assert lib.raw("lh_allreduce_f64")(comm, mine.data_ptr(), 4, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
assert lib.raw("lh_comm_destroy")(comm) == 0
table = gather_rows(rows, 6, world, dev, dist)          # the reference's CSV table across ranks (RCCL all-gather)
if rank == 0:
    print("RESULT " + json.dumps({"agg": agg, "cabi": mine.tolist(), "table": table}))
dist.destroy_process_group()
""" % ROOT


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (the round-end scaling node has 8)")
def test_two_ranks_on_rccl(tmp_path):
    from lookoncetohear_amd import config, synth
    from lookoncetohear_amd.eval import evaluate
    from lookoncetohear_amd.net import Net
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
              "127.0.0.1", "--master-port", "29547"]
    # bench.py launched PLAINLY: it starts its own two ranks under torch.distributed.run
    plain = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=plain, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["metric_sums"][3] == 64.0

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = subprocess.run(launch + [str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    net = Net(**config.TSH_PARAMS).eval()
    net.load_state_dict(config.separator_weights(0), strict=True)
    net = net.to("cuda:0")
    ref, ref_rows = evaluate(net, lambda idx: synth.batch(idx, 16000), n_utts=6, batch_size=2, device="cuda:0")
    assert got["agg"]["n"] == 6 and got["cabi"][3] == 6.0
    assert [t["idx"] for t in got["table"]] == list(range(6))
    for t, q in zip(got["table"], ref_rows):
        assert t["idx"] == q["idx"] and abs(t["si_snr_i"] - q["si_snr_i"]) < 1e-3
    for k in ("si_snr_i", "output_sisnr", "embedding_sim"):
        assert abs(got["agg"][k] - ref[k]) < 1e-4
    assert abs(got["cabi"][0] / 6 - ref["si_snr_i"]) < 1e-3
