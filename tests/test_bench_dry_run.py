"""CPU, world_size 2, gloo: bench.py's own N > 1 plumbing (`--dry-run-cpu`): launched exactly as the driver launches the
scaling runs (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) AND plainly (`python bench.py
--gpus N`: bench.py then starts its own ranks), rank 0 prints ONE JSON line,
the all-reduced metric sums cover both ranks' utterances and equal the single-process sums of the same utterances."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, plain=False):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--dry-run-cpu"]
    if n > 1 and not plain:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
               "127.0.0.1", "--master-port", "29541"] + cmd[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):            # a plain launch is one without a launcher's environment
        env.pop(k, None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_plumbing_world2_gloo():
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.metrics import metric_sums
    two = _run(2)
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["dry_run"] and two["scaling"] == "weak"
    assert two["n_ranks_seen"] == 2 and two["allreduce_32B_us"] > 0          # the group's own size + the exchange step alone
    d = synth.batch([0, 1, 2, 3], 4000)                       # rank 0: utterances 0, 1; rank 1: 2, 3
    ref = metric_sums(0.6 * d["target"] + 0.4 * d["mixture"], d["mixture"], d["target"], d["embedding_gt"][:, 0],
                      d["embedding_gt"][:, 0])
    assert two["metric_sums"][3] == 4.0
    assert torch.allclose(torch.tensor(two["metric_sums"], dtype=torch.float64), ref, rtol=1e-9, atol=1e-9)
    one = _run(1)
    assert one["n_gpus"] == 1 and one["metric_sums"][3] == 2.0 and one["n_ranks_seen"] == 1


def test_bench_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2 --dry-run-cpu` with NO launcher around it (the shape of the driver's N = 1 command with N
    changed; VERDICT r5 weak 3: this died with `AssertionError: --gpus 2 but WORLD_SIZE=1`): bench.py re-executes itself
    under torch.distributed.run and rank 0 prints the same single line as the externally launched form."""
    two = _run(2, plain=True)
    assert two["n_gpus"] == 2 and two["n_ranks_seen"] == 2 and two["metric_sums"][3] == 4.0 and two["dry_run"]
    ext = _run(2)
    assert ext["metric_sums"] == two["metric_sums"]
