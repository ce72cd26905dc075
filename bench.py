#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json on the MI355X-native separator path.

A "step" is one pass of the hot path (`Net.forward`: STFT -> 3 GridNet blocks -> iSTFT, state init included,
exactly what reference src/ts_hear_test.py:138 executes per batch) over one per-GPU batch of synthetic 5 s,
16 kHz binaural mixtures already resident in HBM, followed by the eval loop's metric sums (SI-SNRi etc.,
reference src/ts_hear_test.py:144-146) and — for N > 1 — the one real exchange step of the sharded eval,
an RCCL all-reduce of those 4 sums.  Workload at N=1: BASELINE.json configs[2] ("1xMI355X batch=32 offline
5 s clips"); N>1 shards utterances (weak scaling: 32 per GPU, configs[3] is 8 x 32 = 256).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts the N ranks itself, `self_launch`)

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (largest share of GPU time in the
timed region), measured live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle (a port
of the reference algorithm, oracle/tfgridnet_oracle.py) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_CLIP = 625                 # 5 s @ 16 kHz, hop 128
CLIP_SECONDS = 5.0
FLOPS_PER_CLIP = 46.67e9              # SURVEY.md §8(d) algorithmic FLOPs
BYTES_PER_CLIP = 614.3e6              # SURVEY.md §8(d) algorithmic bytes (fp32, five fused stages per block)
WEIGHT_BYTES = 8.15e6
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense fp16/bf16 MFMA peak (the f16x3 mode executes 3 fp16
                                      # MFMAs per algorithmic fp32 product, so frac <= 1/3 by construction)
PEAK_HBM_GBS = 8000.0

# algorithmic work per LAUNCH and per clip of each kernel (FLOPs, bytes) — SURVEY.md §8(a)/(d), DESIGN.md §4
KERNEL_WORK = {
    "lh_ln_lstm_intra": dict(flops=625 * 97 * 2 * 2 * 128 * 256, bytes=(1 + 2) * 15.52e6, bound="mfma"),
    "lh_ln_lstm_inter": dict(flops=625 * 97 * 2 * 128 * 256, bytes=(1 + 1) * 15.52e6, bound="mfma"),
    # fused kernels: LSTM + output projection + residual (two launches per block for the bidirectional intra path)
    "lh_intra_block": dict(flops=625 * 97 * 2 * (2 * 128 * 256 + 128 * 64), bytes=5 * 15.52e6, bound="mfma"),
    "lh_inter_block": dict(flops=625 * 97 * 2 * (128 * 256 + 64 * 64), bytes=2 * 15.52e6, bound="mfma"),
    # small-batch / streaming variants: same algorithmic work as the kernels they stand in for
    "lh_inter_matvec": dict(flops=625 * 97 * 2 * (128 * 256 + 64 * 64), bytes=2 * 15.52e6, bound="mfma"),
    "lh_intra_stream": dict(flops=625 * 97 * 2 * 2 * 128 * 256, bytes=(1 + 2) * 15.52e6, bound="mfma"),
    "lh_linear_res": dict(flops=625 * 97 * 2 * 96 * 64, bytes=(1.5 + 1 + 1) * 15.52e6, bound="hbm"),   # avg K=96
    "lh_qkv_proj_ln": dict(flops=625 * 97 * 2 * 64 * 112, bytes=15.52e6 + 2 * 5.82e6 + 15.52e6, bound="hbm"),
    "lh_local_attn": dict(flops=4 * 625 * 50 * 2 * (582 + 1552), bytes=2 * 5.82e6 + 2 * 15.52e6, bound="hbm"),
    "lh_proj_ln_res": dict(flops=625 * 97 * 2 * 64 * 64, bytes=3 * 15.52e6, bound="hbm"),
    "lh_stft_conv_in": dict(flops=0.37e9, bytes=16.16e6, bound="hbm"),
    "lh_deconv_istft": dict(flops=0.37e9, bytes=16.16e6, bound="hbm"),
    "lh_embed_proj_ln": dict(flops=3.2e6, bytes=6.4e6, bound="hbm"),
}


# kernel launches behind one C-ABI call (HIP events bracket the call; rocprofv3 reports per kernel launch)
LAUNCHES_PER_CALL = {"lh_intra_block": 2, "lh_embed_proj_ln": 2, "lh_metric_sums": 2}
# the kernel function behind each call, as rocprofv3 names it (profiles/*kernel_stats*.csv)
KERNEL_NAME = {"lh_inter_matvec": "k_inter_matvec", "lh_intra_stream": "k_intra_stream",
               "lh_intra_block": "k_intra_xp", "lh_inter_block": "k_inter_xp",
               "lh_local_attn": "k_local_attn", "lh_qkv_proj_ln": "k_qkv_proj_ln", "lh_proj_ln_res": "k_proj_ln_res",
               "lh_deconv_istft": "k_deconv_istft", "lh_stft_conv_in": "k_stft_conv_in"}


# enrollment embedder (T = 1251 frames x 65 bins x 64 channels per clip; A_e = one fp32 activation tensor = 20.8 MB):
# algorithmic HBM bytes per C-ABI call and clip with one round trip between the stages of a call —
#   axis call: read x, write hidden states h (2 directions x 64 = 2 A_e), read h, read x (residual), write x      = 6 A_e
#   attention block: read x, write + read Q, K (2 x 4 heads x 1251 x 520 fp32 = 10.4 MB each... as split fp16: same bytes),
#                    V (A_e), write + read the merged heads (A_e), read x (residual), write x                    = 6 A_e + 4 QK
#   (the materialised score matrix is NOT algorithmic: it is the traffic the roofline.traffic figure exposes)
EMBED_AE = 1251 * 65 * 64 * 4.0
EMBED_QK = 4 * 1251 * 520 * 4.0
EMBED_CALL_BYTES = {"lh_emb_axis.intra": 6 * EMBED_AE, "lh_emb_axis.inter": 6 * EMBED_AE,
                    "lh_emb_attn_block": 6 * EMBED_AE + 4 * EMBED_QK, "lh_emb_head": EMBED_AE,
                    "lh_emb_frontend": 2 * EMBED_AE}

PACKAGE_POWER_CAP_W = 1400.0          # MI355X package power limit (rocm-smi --showmaxpower on the bench boxes)
MEASURED_MFMA16_TFLOPS_AT_CAP = 2310.0   # pure v_mfma_f32_16x16x32_f16 stream, every CU, non-trivial operands: 1280 W at
                                          # 2.33 GHz (profiles/r03a_power_per_instruction.txt); 32x32x16: 2000 at 1.91 GHz


def power_leg(step_fn, seconds=2.0):
    """Package power / shader clock while `step_fn` runs back to back for `seconds` (rocm-smi sampled from a second
    thread; the timed region of the bench is far too short for a sample).  The whole path runs at the package power
    limit, so joules per step — not cycles — is what bounds it."""
    import re
    import subprocess
    import threading
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
                sc = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", o)
                if pw:
                    samples.append((float(pw.group(1)), int(sc.group(1)) if sc else 0))
            except Exception:            # no rocm-smi on this box: the leg reports null
                return

    try:
        for _ in range(3):
            step_fn()
        torch.cuda.synchronize()
        th = threading.Thread(target=sampler)
        th.start()
        n, t0 = 0, time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                step_fn()
            n += 10
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop[0] = True
        th.join()
        s_ = samples[1:] or samples
        if not s_:
            return None
        ms = e0.elapsed_time(e1) / n
        pw = sum(a for a, _ in s_) / len(s_)
        sc = sum(b for _, b in s_) / len(s_)
        return {"package_w_avg": pw, "package_w_max": max(a for a, _ in s_), "package_cap_w": PACKAGE_POWER_CAP_W,
                "frac_of_cap": pw / PACKAGE_POWER_CAP_W, "sclk_mhz_avg": sc, "samples": len(s_), "ms_per_step": ms,
                "joules_per_step": pw * ms * 1e-3, "seconds": seconds,
                "note": "steps run back to back for `seconds`, rocm-smi sampled concurrently; per-call figures: "
                        "profiles/r03*_power_by_call*.txt"}
    except Exception as e:               # never lose the headline line to the side measurement
        stop[0] = True
        return {"error": repr(e)[:200]}


def limited_by(power, roof):
    """What the dominant kernel is held by, derived from THIS run: the `power` leg (package power of back-to-back steps
    against the cap) decides between "package power" and the counter-derived bound (busy fractions of the committed PMC
    passes, profiles/pmc_traffic.json).  No literal figures: a run below the cap does not claim it."""
    busy = ", ".join(f"{k} {roof[k]:.2f}" for k in ("mfma_busy", "valu_busy") if k in roof)
    if not power or "package_w_avg" not in power:
        return "not measured in this run (no power leg)" + (f"; PMC: {busy}" if busy else "")
    f = power["frac_of_cap"]
    head = (f"whole forward at {power['package_w_avg']:.0f} W = {f:.2f} of the {power['package_cap_w']:.0f} W package cap, "
            f"sclk {power['sclk_mhz_avg']:.0f} MHz (this run, back-to-back steps)")
    if f >= 0.97:
        return "package power: " + head
    if busy:
        return (f"below the package cap in this run ({head}); the dominant kernel is issue / dependency-bound: {busy} "
                f"(profiles/pmc_traffic.json); per-call power: profiles/*power_by_call*.txt")
    return f"below the package cap in this run ({head})"


def cpu_baseline(sample_clips=4, repeats=2, budget_s=120.0, dump=None, embed_rows=1):
    """The reference's CPU path on this box's host cores: `oracle/aten_port.py` issues the reference's own ATen operator
    sequence (nn.LSTM's aten::lstm, unfold(2, 50, 1) + reshape copy, matmul, softmax ...; bit-identical to the unmodified
    reference where that can be imported, `python -m oracle.aten_port`), on a bounded sample of the same workload: one
    micro-batch of 4 x 5 s utterances — the reference's own eval batch size (src/ts_hear_test.py:121); 32 in one call
    would need ~25 GB of unfold temporaries per block.  Timed with 32 and with 64 threads (never all cores of a large
    box, see below); the better one is `value`, both are in `frames_per_s_by_threads`.

    `dump` (a path): the leg's OUTPUTS are kept instead of thrown away — the 4 reference waveforms,
    and the embedder oracle's embeddings of `embed_rows` full-length enrollments (fp32 torch CPU, timed as the embedder's
    cpu_baseline) — so the main process can hold the HIP outputs of the SAME clips against them (`parity` in the line)."""
    from lookoncetohear_amd import synth, config
    from oracle import aten_port as P
    host = len(os.sched_getaffinity(0))
    d = P.Dims(config.TSH_PARAMS)
    sd = config.separator_weights(0)
    b = synth.batch(list(range(sample_clips)), 80000)
    runs = {}
    y_ref = None
    t_start = time.perf_counter()
    # 32 and 64 threads: the step-serial LSTM / softmax ops stop scaling long before a whole socket, and torch's all-core
    # run of this op mix does not even finish its 1 s warm-up clip in 190 s on the 256-core GPU box (it spends its time
    # in OpenMP barriers), which is how the round-1/2 drivers' bench lines lost their cpu_baseline to the time limit.
    say = lambda m: print(f"[cpu_baseline {time.perf_counter() - t_start:6.1f} s] {m}", file=sys.stderr, flush=True)
    say(f"weights + inputs ready, host cores {host}")
    for threads in sorted({min(host, 32), min(host, 64)}):
        if runs and time.perf_counter() - t_start > budget_s / 3:
            break
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        P.forward(d, sd, b["mixture"][:1, :, :16000], b["embedding_gt"][:1])     # warm-up (1 s clip)
        warm = time.perf_counter() - t0
        say(f"{threads} threads: warm-up (1 s clip) {warm:.1f} s")
        if runs and warm * 5 * sample_clips > budget_s / 3:                       # would not fit: keep what we have
            break
        best = float("inf")
        for _ in range(repeats):
            t0 = time.perf_counter()
            y_ref = P.forward(d, sd, b["mixture"], b["embedding_gt"])
            best = min(best, time.perf_counter() - t0)
            say(f"{threads} threads: pass {best:.1f} s")
            if time.perf_counter() - t_start > budget_s:
                break
        runs[threads] = best
    cores = min(runs, key=runs.get)
    best = runs[cores]
    out = dict(value=sample_clips * FRAMES_PER_CLIP / best, unit="frames/s", cores=cores, kind="port",
               rtf=best / (sample_clips * CLIP_SECONDS), host_cores=host,
               frames_per_s_by_threads={str(k): sample_clips * FRAMES_PER_CLIP / v for k, v in runs.items()},
               sample=f"{sample_clips} x 5 s clips in one micro-batch (the reference's eval batch), best of {repeats}; "
                      f"oracle/aten_port.py = the reference's ATen op sequence (bit-identical to the reference in the "
                      f"build container), torch CPU fp32; threads tried: {sorted(runs)} of {host} host cores")
    if dump:
        import numpy as np
        emb_ref, emb_cpu = None, None
        try:                                                   # embedder leg: never lose the separator numbers to it
            emb_ref, emb_cpu = embed_cpu_leg(b["mixture"][:embed_rows], min(host, 32))
            say(f"embedder oracle: {emb_cpu['seconds']:.1f} s for {embed_rows} x 5 s")
        except Exception as e:
            emb_cpu = {"value": None, "sample": "failed: " + repr(e)[:200]}
        np.savez(dump, y=y_ref.numpy(), **({"emb": emb_ref.numpy()} if emb_ref is not None else {}))
        out["embed_cpu_baseline"] = emb_cpu
    return out


def embed_cpu_leg(x, threads):
    """cpu_baseline leg of the enrollment embedder (BASELINE configs[4]): oracle/embedder_oracle.py (torch CPU fp32) on the
    full-length enrollments `x` [n, 2, 80000].  Returns (embeddings [n, 256], the cpu_baseline object)."""
    from oracle import embedder_oracle as E
    torch.set_num_threads(threads)
    cfg = E.ECfg(**E.EMBED_PARAMS)
    sd = E.synthetic_state_dict(cfg, 0)
    E.forward(cfg, sd, x[:1, :, :8000])                        # warm-up (0.5 s clip)
    t0 = time.perf_counter()
    emb = E.forward(cfg, sd, x)
    dt = time.perf_counter() - t0
    n, T = x.shape[0], x.shape[-1] // 64 + 1
    return emb, {"value": n * T / dt, "unit": "frames/s", "cores": threads, "kind": "port", "seconds": dt,
                 "clips_per_s": n / dt,
                 "sample": f"oracle/embedder_oracle.py (torch CPU fp32; trunk restated from espnet2, parity unpinned), "
                           f"{n} x 5 s enrollments = {n * T} frames, one pass"}


def cpu_baseline_subprocess(timeout_s=300, dump=None):
    """Runs cpu_baseline() in a child process with a hard time limit so the GPU measurement can never hang on it."""
    import subprocess
    code = f"import json, bench; print('CPUBASE ' + json.dumps(bench.cpu_baseline(dump={dump!r})))"
    try:
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        for line in out.stdout.splitlines():
            if line.startswith("CPUBASE "):
                return json.loads(line[len("CPUBASE "):])
        return dict(value=None, unit="frames/s", cores=0, kind="port", sample="failed: " + out.stderr[-300:])
    except subprocess.TimeoutExpired as e:
        tail = (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))[-400:]
        return dict(value=None, unit="frames/s", cores=0, kind="port", sample=f"timed out after {timeout_s} s; progress: {tail}")


def parity_object(net, dev, dump, sample_clips=4):
    """`parity` of the bench line: the 4 clips the cpu_baseline leg just ran
    through the reference's ATen sequence, through the HIP net in THIS process — waveform max-abs and the SI-SNRi
    difference per utterance (north star: <= 1e-3, <= 0.05 dB).  The oracle is only the checker here."""
    import numpy as np
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.metrics import per_utterance
    ref = np.load(dump)
    y_ref = torch.from_numpy(ref["y"])
    b = synth.batch(list(range(sample_clips)), 80000)
    with torch.no_grad():
        y = net(b["mixture"].to(dev), b["embedding_gt"].to(dev))
        bad = net.range_status(dev)
    y = y.cpu()
    e = b["embedding_gt"][:, 0]
    _, si_h, _ = per_utterance(y.double(), b["mixture"].double(), b["target"].double(), e, e)
    _, si_r, _ = per_utterance(y_ref.double(), b["mixture"].double(), b["target"].double(), e, e)
    return {"max_abs": float((y - y_ref).abs().max()), "d_sisnri_db": float((si_h - si_r).abs().max()),
            "clips": sample_clips, "out_amp": float(y_ref.abs().max()), "range_flag_raised": bool(bad),
            "against": "reference ATen sequence fp32 (oracle/aten_port.py, the outputs of this run's cpu_baseline leg)",
            "tolerance": {"max_abs": 1e-3, "d_sisnri_db": 0.05},
            "ok": bool(float((y - y_ref).abs().max()) <= 1e-3 and float((si_h - si_r).abs().max()) <= 0.05 and not bad)}, ref


def secondary_parity(sec, kept, ref):
    """`parity` objects of the secondary legs: the waveforms those legs kept (`kept`: rows 0..3 of
    the B = 256 batch, row 0 of B = 1, the streamed clip, rows 0..3 of the second in-flight replica — all utterances 0..3) against
    the reference outputs of the cpu_baseline leg (`ref['y']`, the reference's ATen sequence on the same four clips)."""
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.metrics import per_utterance
    y_ref = torch.from_numpy(ref["y"])
    b = synth.batch(list(range(y_ref.shape[0])), 80000)
    e = b["embedding_gt"][:, 0]

    def obj(y, n_samples=None):
        k = y.shape[0]
        r, m, t = y_ref[:k], b["mixture"][:k], b["target"][:k]
        if n_samples is not None:
            r, m, t = r[..., :n_samples], m[..., :n_samples], t[..., :n_samples]
        _, si_h, _ = per_utterance(y.double(), m.double(), t.double(), e[:k], e[:k])
        _, si_r, _ = per_utterance(r.double(), m.double(), t.double(), e[:k], e[:k])
        ma, ds = float((y - r).abs().max()), float((si_h - si_r).abs().max())
        return {"max_abs": ma, "d_sisnri_db": ds, "clips": k, "against": "reference ATen sequence fp32 (this run's cpu_baseline leg)",
                "tolerance": {"max_abs": 1e-3, "d_sisnri_db": 0.05}, "ok": bool(ma <= 1e-3 and ds <= 0.05)}

    for key, leg in (("_y_offline_b1", "offline_b1"), ("_y_offline_b4", "offline_b4"), ("_y_offline_b256", "offline_b256"),
                     ("_y_two_in_flight", "offline_b32_two_in_flight")):
        if key in kept and isinstance(sec.get(leg), dict):
            sec[leg]["parity"] = obj(kept[key])
    if "_y_stream_b1" in kept and isinstance(sec.get("stream_b1"), dict):
        y = kept["_y_stream_b1"]
        sec["stream_b1"].setdefault("parity", {}).update(
            {("vs_reference_" + k if k in ("max_abs", "d_sisnri_db", "ok") else k): v for k, v in obj(y, y.shape[-1]).items()
             if k not in ("clips", "tolerance")})
        sec["stream_b1"]["parity"]["tolerance_vs_reference"] = {"max_abs": 1e-3, "d_sisnri_db": 0.05}
        sec["stream_b1"]["parity"]["samples"] = int(y.shape[-1])


def gpu_library_baseline(sample_clips=4, repeats=3):
    """Context only, not the product and not credit: the reference's own operator sequence
    (`oracle/aten_port.py`, unchanged) on THIS GPU through PyTorch-ROCm's libraries (MIOpen LSTM, rocBLAS / hipBLASLt
    matmul, ATen elementwise) — what `python -m src.ts_hear_test --device cuda` (reference src/ts_hear_test.py:175) would
    run — on micro-batches of 4 x 5 s (the reference's eval batch; its unfold copies are ~4.3 GB per block at that size)."""
    from lookoncetohear_amd import synth, config
    from oracle import aten_port as P
    dev = torch.device("cuda", 0)
    d = P.Dims(config.TSH_PARAMS)
    sd = {k: v.to(dev) for k, v in config.separator_weights(0).items()}
    b = synth.batch(list(range(sample_clips)), 80000)
    mix, emb = b["mixture"].to(dev), b["embedding_gt"].to(dev)
    P.forward(d, sd, mix[:1, :, :16000], emb[:1])
    y = P.forward(d, sd, mix, emb)
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        y = P.forward(d, sd, mix, emb)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return dict(value=sample_clips * FRAMES_PER_CLIP / best, unit="frames/s", ms_per_micro_batch=best * 1e3,
                rtf=best / (sample_clips * CLIP_SECONDS), kind="reference ATen op sequence on PyTorch-ROCm libraries",
                sample=f"{sample_clips} x 5 s clips per call (the reference's eval batch), best of {repeats}, fp32, eager",
                checksum=float(y.double().abs().sum()))


def gpu_library_baseline_subprocess(timeout_s=240):
    """In a child process with a hard limit: MIOpen may JIT-compile its LSTM kernels on first use."""
    import subprocess
    code = "import json, bench; print('GPULIB ' + json.dumps(bench.gpu_library_baseline()))"
    try:
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        for line in out.stdout.splitlines():
            if line.startswith("GPULIB "):
                return json.loads(line[len("GPULIB "):])
        return dict(value=None, unit="frames/s", sample="failed: " + out.stderr[-300:])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit="frames/s", sample=f"timed out after {timeout_s} s")


def bench_stream(args, net, dev, rank, world):
    """BASELINE configs[1]: batch-1 (or --batch N independent streams) chunked forward, hop 128, 64-sample look-ahead.
    A step = one 8 ms chunk through the graph-captured per-chunk launch sequence; replicas only across GPUs."""
    from lookoncetohear_amd import synth
    B = 1 if args.batch == 32 else args.batch
    d = synth.batch(list(range(B)), 80000)
    mix = torch.nn.functional.pad(d["mixture"], (0, 64)).to(dev)
    st = net.make_streamer(B, dev, use_graph=True)
    st.set_embedding(d["embedding_gt"].to(dev))
    nchunks = 625
    chunks = [mix[:, :, i * 128:i * 128 + 192].contiguous() for i in range(nchunks)]
    for i in range(args.warmup):
        st.step(chunks[i % nchunks])
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        t1 = time.perf_counter()
        y = st.step(chunks[i % nchunks])
        torch.cuda.synchronize()                 # a real-time consumer needs the chunk before the next one arrives
        lat.append(time.perf_counter() - t1)
    elapsed = time.perf_counter() - t0
    lat.sort()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        print(json.dumps({
            "metric": "streaming chunk latency / real-time factor (128-sample hop, 64-sample look-ahead)",
            "value": B * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "replicas",
            "vs_baseline": None, "dtype": "f32 via f16x3 split (3x fp16 MFMA per product, fp32 accumulate)", "data": "synthetic", "rtf": ms / 8.0,
            "latency_ms": {"p50": lat[len(lat) // 2] * 1e3, "p99": lat[int(len(lat) * 0.99)] * 1e3, "max": lat[-1] * 1e3},
            "config": {"workload": f"BASELINE configs[1]: {B} stream(s), 8 ms chunks with carried state, HIP-graph replay",
                       "batch_per_gpu": B, "gemm_mode": net.gemm_mode}}))


def embed_instrumented_step(net, x):
    """One forward of the embedder on ONE stream with HIP events around every C-ABI call -> {call: launches, total_ms, avg_ms}."""
    ns_keep, net.n_streams, net._prof = net.n_streams, 1, []
    try:
        net(x)
        torch.cuda.synchronize()
        prof = net._prof
    finally:
        net._prof, net.n_streams = None, ns_keep
    per = {}
    for name, e0, e1 in prof:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    return {k: dict(launches=len(v), total_ms=sum(v), avg_ms=sum(v) / len(v)) for k, v in per.items()}


def embed_roofline(kern, B, T=1251):
    """`roofline` of the embedder's dominant C-ABI call (largest share of one instrumented step): algorithmic
    fp32-equivalent FLOPs / bytes of the call (per clip x B) over its HIP-event duration; `traffic` and the busy fractions
    from the committed PMC passes of `bench.py --mode embed` (profiles/pmc_traffic.json, section `embed`)."""
    work = {      # algorithmic fp32-equivalent FLOPs per clip (T = 1251 frames, 65 bins)
        "lh_emb_attn_block": 2.0 * T * 65 * 64 * (128 + 64) + 4 * (2.0 * T * T * 520 + 2.0 * T * T * 1040),
        "lh_emb_axis.intra": 2.0 * T * 62 * (256 * 512 + 128 * 512) + 2.0 * T * 65 * 512 * 64,
        "lh_emb_axis.inter": 2.0 * 65 * (T - 3) * (256 * 512 + 128 * 512) + 2.0 * T * 65 * 512 * 64,
        "lh_emb_head": 2.0 * T * 4160 * 256,
        "lh_emb_frontend": 2.0 * T * 2 * 128 * 130 + 2.0 * T * 65 * 36 * 64,
    }
    dom = max(kern, key=lambda k: kern[k]["total_ms"])
    ach = work[dom] * B / (kern[dom]["avg_ms"] * 1e-3) / 1e12
    exact = dom == "lh_emb_frontend"
    peak = PEAK_FP32_MFMA_TFLOPS if exact else PEAK_F16_MFMA_TFLOPS
    traffic, tsrc, busy = None, None, {}
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")        # HBM bytes per call from the committed PMC passes
    if os.path.exists(tfile):
        te = json.load(open(tfile)).get("embed", {})
        if te.get("batch_per_gpu") == B and dom in te.get("calls", {}):
            tc = te["calls"][dom]
            traffic = tc.get("hbm_bytes_per_call")
            busy = {k: tc[k] for k in ("mfma_busy", "valu_busy", "dominant_kernel", "dominant_kernel_avg_ms",
                                       "dominant_kernel_hbm_bytes_per_launch") if k in tc}
            tsrc = ("profiles/pmc_traffic.json `embed` section (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                    "`bench.py --mode embed`, summed over the kernels behind the call; source: %s) - not re-measured in this run"
                    % te.get("source"))
    byts = EMBED_CALL_BYTES.get(dom)
    return {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": kern[dom]["avg_ms"],
            "launches_per_step": kern[dom]["launches"],
            "share_of_gpu_time": kern[dom]["total_ms"] / sum(v["total_ms"] for v in kern.values()),
            "algorithmic_bytes_per_call": byts * B if byts else None,
            "frac_hbm": (byts * B / (kern[dom]["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS) if byts else None, **busy,
            "note": "one C-ABI call = several kernels; algorithmic fp32-equivalent FLOPs" +
                    ("" if exact else "; split-precision fp16 MFMA (3 per product), peak = dense fp16")}


def bench_embed(args, dev, rank, world, dist):
    """BASELINE configs[4]: the enrollment d-vector embedder (configs/embed.json) on 64 x 5 s binaural clips per GPU.
    A step = one forward of the batch; utterances shard across ranks with no exchange at all (embeddings stay local)."""
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.embed_net import EmbedTFGridNet
    from lookoncetohear_amd import config
    B = args.embed_batch if args.batch == 32 else args.batch
    net = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
    net.load_state_dict(config.embedder_weights(0), strict=True)
    net = net.to(dev)
    uniq = min(B, 8)
    x = synth.batch([rank * B + i for i in range(uniq)], 80000)["mixture"]
    x = x.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous().to(dev)
    T = 80000 // 64 + 1
    with torch.no_grad():
        for _ in range(args.warmup):
            net(x)
        torch.cuda.synchronize()
        # one instrumented step on ONE stream (HIP events around every C-ABI call): the per-call breakdown and the dominant
        # call; the timed region then runs the product configuration (two half-batches on two streams by default)
        kern = embed_instrumented_step(net, x)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            emb = net(x)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if rank != 0:
        return
    roof = embed_roofline(kern, B)
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        # cpu_baseline leg (the only use of oracle/ here): two full-length enrollments through the CPU oracle, timed; the
        # same two rows of the HIP batch are then held against them (north star: "embedding cosine match vs CPU ref")
        ref, cpu = embed_cpu_leg(x[:2].cpu(), min(32, len(os.sched_getaffinity(0))))
        cos = torch.nn.functional.cosine_similarity(emb[:2].cpu().double(), ref.double(), dim=-1)
        parity = {"embedding_cos_min": float(cos.min()), "max_abs": float((emb[:2].cpu().double() - ref.double()).abs().max()),
                  "clips": 2, "against": "oracle/embedder_oracle.py fp32, full-length enrollments (trunk restated from espnet2: "
                                         "parity unpinned; front end + head pinned to reference code, DESIGN.md §2)"}
    print(json.dumps({
        "metric": "enrollment embedder frames_per_sec (5 s 16 kHz binaural clips, 1251 STFT frames each)",
        "value": world * B * args.steps * T / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 via f16x3 split (3x fp16 MFMA per product, fp32 accumulate)", "data": "synthetic", "clips_per_sec": world * B * args.steps / elapsed,
        "config": {"workload": f"BASELINE configs[4]: configs/embed.json d-vector embedder, {B} x 5 s clips per GPU "
                               "(random-init weights; oracle parity unpinned, see DESIGN.md)", "batch_per_gpu": B},
        "roofline": roof,
        "cpu_baseline": cpu, "parity": parity, "kernels_ms_per_step": {k: v["total_ms"] for k, v in kern.items()},
        "kernels_note": f"per-call HIP-event times of one instrumented single-stream step; the timed region ran {net.n_streams} "
                        f"half-batch(es) on {net.n_streams} HIP stream(s)", "n_streams": net.n_streams,
        "embedding_norm_mean": float(emb.norm(dim=1).mean())}))


def bench_render(args, dev, rank, world, dist):
    """binaural rendering of `batch` utterances x (3 sources + noise) x 5 s with `--rir-len`-tap
    2-ear impulse responses (256 = HRIR-length at 16 kHz, 4096 = BRIR-length).  A step = render + mix of the batch."""
    import numpy as np
    from lookoncetohear_amd.render import BinauralRenderer
    from oracle import render_oracle as R
    B = 64 if args.batch == 32 else args.batch
    N, Lh, S1 = 80000, args.rir_len, 4
    sc = [R.synthetic_scene(rank * 8 + i, N, 3, Lh, reverb=Lh > 1024) for i in range(min(B, 8))]
    rep = (B + len(sc) - 1) // len(sc)
    srcs = torch.from_numpy(np.stack([s[0] for s in sc])).repeat(rep, 1, 1)[:B].contiguous().to(dev)
    rirs = torch.from_numpy(np.stack([s[1] for s in sc])).repeat(rep, 1, 1, 1)[:B].contiguous().to(dev)
    gains = torch.ones(B, S1, device=dev)
    gains[:, 3] = torch.tensor([s[2] for s in sc]).repeat(rep)[:B].to(dev)
    tgt = torch.tensor([s[3] for s in sc]).repeat(rep)[:B].to(dev)
    rr = BinauralRenderer()
    for _ in range(args.warmup):
        rr.render(srcs, rirs, gains, tgt)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        out = rr.render(srcs, rirs, gains, tgt)
    e1.record()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if rank != 0:
        return
    call_ms = e0.elapsed_time(e1) / args.steps
    flops = 2.0 * N * Lh * S1 * 2 * B                       # FIR MACs x 2 (upper bound: ignores the start-up triangle)
    bytes_ = (S1 * N + S1 * 2 * N * 3 + 2 * 2 * N) * 4.0 * B  # src in, events out + 2 reads by the mix, mixture + target out
    valu = flops / (call_ms * 1e-3) / 1e12
    hbm = bytes_ / (call_ms * 1e-3) / 1e9
    fft_path = 1024 <= Lh <= 4097                           # lh_render.hip: overlap-save FFT convolution (1/16 of the direct FLOPs)
    compute_bound = (not fft_path) and valu / PEAK_FP32_MFMA_TFLOPS > hbm / PEAK_HBM_GBS
    cpu = None
    if not args.no_cpu_baseline:
        t1 = time.perf_counter()
        R.render(*sc[0])
        dt = time.perf_counter() - t1
        cpu = {"value": 1.0 / dt, "unit": "utterances/s", "cores": 1, "kind": "port",
               "sample": "oracle/render_oracle.py = the reference's scipy.signal.convolve calls, 1 utterance (4 rows x 2 ears)"}
    print(json.dumps({
        "metric": "binaural rendering utterances_per_sec (3 sources + noise, 5 s 16 kHz, 2 ears)",
        "value": world * B * args.steps / elapsed, "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SURVEY 8f-3: {B} utterances x 4 rows x 80000 samples, {Lh}-tap 2-ear responses", "batch_per_gpu": B},
        "roofline": ({"kernel": "lh_render_binaural (k_fir_causal dominant)", "bound": "mfma", "achieved": valu,
                      "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": valu / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                      "note": "fp32 vector FMA (v_pk_fma_f32): same 157 TFLOP/s peak as the fp32 MFMA; not a matrix-core kernel"}
                     if compute_bound else
                     {"kernel": "lh_render_binaural" + (" (k_fft_conv dominant)" if fft_path else ""), "bound": "hbm",
                      "achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS, "traffic": None,
                      **({"direct_form_equivalent_tflops": valu} if fft_path else {})}),
        "avg_call_ms": call_ms, "cpu_baseline": cpu, "norm_factor_mean": float(out[2].mean())}))


def timed_region(step, steps, dist, dev, sync):
    """The contract's timed region: barrier + device sync on both sides of exactly `steps` calls of `step`, MAX over
    ranks.  Shared by the GPU path (RCCL) and the CPU dry run of the plumbing (gloo)."""
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), elapsed, out


def dry_run_cpu(args, rank, world):
    """`--dry-run-cpu`: the N > 1 plumbing of this file — process group, utterance sharding by rank, the 32-byte
    all-reduce of the metric sums inside the step, barrier-bracketed timing with MAX over ranks, rank 0's JSON line — on
    the gloo backend with host tensors and NO separator (outputs = a fixed mix of target and mixture), so the
    distributed path can be exercised where there is no GPU (tests/test_bench_dry_run.py).  Not a measurement."""
    import torch.distributed as dist
    from lookoncetohear_amd import synth
    from lookoncetohear_amd.metrics import metric_sums
    if world > 1:
        dist.init_process_group(backend="gloo")
    B = min(args.batch, 2)
    d = synth.batch([rank * B + i for i in range(B)], 4000)
    out = 0.6 * d["target"] + 0.4 * d["mixture"]

    def step():
        sums = metric_sums(out, d["mixture"], d["target"], d["embedding_gt"][:, 0], d["embedding_gt"][:, 0])
        if world > 1:
            dist.all_reduce(sums)
        return sums

    for _ in range(args.warmup):
        step()
    elapsed, _, sums = timed_region(step, args.steps, dist if world > 1 else None, "cpu", lambda: None)
    ranks_seen, allreduce_us = world, None
    if world > 1:                        # same fields as the GPU line: the group's own size, the exchange step by itself
        ranks_seen = dist.get_world_size()
        buf = sums.clone()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        allreduce_us = (time.perf_counter() - t0) / 20 * 1e6
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (gloo, no separator): plumbing only", "value": world * B * args.steps / elapsed,
                          "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "plumbing dry run", "batch_per_gpu": B, "global_batch": world * B},
                          "n_ranks_seen": ranks_seen, "allreduce_32B_us": allreduce_us,
                          "metric_sums": [float(v) for v in sums.tolist()]}))
    if world > 1:
        dist.destroy_process_group()


def _time_forward(fn, steps, warmup):
    """ms per call of `fn` (device-synchronised wall clock around `steps` calls)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def secondary_measurements(net, dev, mix8, emb8):
    """The other configurations BASELINE.json's north_star names, from the same process and the same weights, a few
    steps each (N = 1 only): offline batch 1 and batch 256, batch-1 streaming (configs[1]), the enrollment embedder
    (configs[4]), and the headline batch once more with the exact-fp32 recurrences for contrast.  `mix8` / `emb8`: the
    8 resident utterances the headline batch is tiled from."""
    out = {}
    log = lambda m: print(f"[bench secondary] {m}", file=sys.stderr, flush=True)
    with torch.no_grad():
        for B in (1, 4, 256):
            try:
                mix = mix8.repeat((B + 7) // 8, 1, 1)[:B].contiguous()
                emb = emb8.repeat((B + 7) // 8, 1, 1)[:B].contiguous()
                ms = _time_forward(lambda: net(mix, emb), 50 if B <= 4 else 3, 5 if B <= 4 else 2)
                # rows 0..3 (B = 256) / row 0 (B = 1) = utterances 0..3, the clips of the cpu_baseline leg: held against its
                # reference outputs once that leg has run (`secondary_parity`)
                out[f"_y_offline_b{B}"] = net(mix, emb)[:4].cpu()
                out[f"offline_b{B}"] = {"ms_per_step": ms, "frames_per_s": B * FRAMES_PER_CLIP / ms * 1e3,
                                        "rtf": ms * 1e-3 / (B * CLIP_SECONDS),
                                        "time_windows": net._n_time_chunks(B, FRAMES_PER_CLIP, 1),
                                        "workload": f"{B} x 5 s clips, offline forward" +
                                                    (" (BASELINE configs[3]'s global batch on one GPU)" if B == 256 else "") +
                                                    (" (the reference's eval batch, src/ts_hear_test.py:121)" if B == 4 else "")}
                log(f"offline B={B}: {ms:.3f} ms")
                del mix, emb
            except Exception as e:          # e.g. out of memory on a smaller part: report, do not lose the headline line
                out[f"offline_b{B}"] = {"error": repr(e)[:200]}
            net._ws.clear()
            torch.cuda.empty_cache()
        # exact-fp32 recurrences (v_mfma_f32_16x16x4_f32) at the headline batch
        try:
            B = 32
            mix = mix8.repeat(4, 1, 1).contiguous()
            emb = emb8.repeat(4, 1, 1).contiguous()
            net.gemm_mode = "f32rec"
            ms = _time_forward(lambda: net(mix, emb), 2, 1)
            out["offline_b32_f32rec"] = {"ms_per_step": ms, "frames_per_s": B * FRAMES_PER_CLIP / ms * 1e3,
                                         "workload": "headline batch with gemm_mode='f32rec': exact fp32 MFMA in the two "
                                                     "RECURRENCES only; the five frame kernels stay split-precision (an all-fp32 "
                                                     "run is gemm_mode='f32all', test-only reference kernels, not timed here)"}
            log(f"offline B=32 f32rec: {ms:.3f} ms")
        except Exception as e:
            out["offline_b32_f32rec"] = {"error": repr(e)[:200]}
        finally:
            net.gemm_mode = "f16x3"
            net._ws.clear()
            torch.cuda.empty_cache()
        # two batches in flight (context for DESIGN.md §11, NOT the headline): two Net replicas with the same weights on two HIP
        # streams, steps alternating — what the CUs the inter LSTM leaves idle (194 of 256 busy) are worth to a caller that has
        # the next batch ready.  ms = wall / batches: an inverse throughput, not a latency.
        try:
            from lookoncetohear_amd import config
            from lookoncetohear_amd.net import Net
            B = 32
            mix = mix8.repeat(4, 1, 1).contiguous()
            emb = emb8.repeat(4, 1, 1).contiguous()
            net2 = Net(**config.TSH_PARAMS).eval()
            net2.load_state_dict(net.state_dict(), strict=True)
            net2 = net2.to(dev)
            # (the Net's own pool of side streams — the ones `time_chunks_small` uses — instead of two more: past four streams per
            # process the hardware queues are shared and the legs below would depend on which stream landed where)
            nets, streams = [net, net2], net._lanes(dev, 3).streams[1:3]
            cur = torch.cuda.current_stream(dev)

            def in_flight(n):
                for s_ in streams:
                    s_.wait_stream(cur)
                for i in range(n):
                    with torch.cuda.stream(streams[i & 1]):
                        nets[i & 1](mix, emb)
                for s_ in streams:
                    cur.wait_stream(s_)

            in_flight(4)
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[1]):
                y2 = nets[1](mix, emb)
            torch.cuda.synchronize()
            y1 = net(mix, emb)
            same = bool(torch.equal(y1, y2))
            out["_y_two_in_flight"] = y2[:4].cpu()
            del y1, y2
            ms1 = _time_forward(lambda: net(mix, emb), 8, 2)
            t0 = time.perf_counter()
            in_flight(16)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) / 16 * 1e3
            out["offline_b32_two_in_flight"] = {"ms_per_batch": ms2, "ms_per_batch_one_in_flight": ms1, "ratio": ms2 / ms1,
                                                "frames_per_s": B * FRAMES_PER_CLIP / ms2 * 1e3,
                                                "replica_bit_identical_to_headline_net": same,
                                                "workload": "two batch-32 forwards in flight (two Net replicas, two HIP streams, "
                                                            "batches alternating); the headline keeps ONE in flight"}
            log(f"offline B=32, two in flight: {ms2:.3f} ms per batch (one in flight, same loop: {ms1:.3f})")
            del net2, nets, mix, emb
        except Exception as e:
            out["offline_b32_two_in_flight"] = {"error": repr(e)[:200]}
        net._ws.clear()
        torch.cuda.empty_cache()
        # streaming, BASELINE configs[1]: one stream, 8 ms chunks, HIP-graph replay; latency per chunk incl. the sync a
        # real-time consumer needs
        try:
            st = net.make_streamer(1, dev, use_graph=True)
            st.set_embedding(emb8[:1, 0])
            mixp = torch.nn.functional.pad(mix8[:1], (0, 64))
            chunks = [mixp[:, :, i * 128:i * 128 + 192].contiguous() for i in range(625)]
            ys = []
            for i in range(20):
                ys.append(st.step(chunks[i]).clone())
            torch.cuda.synchronize()
            lat = []
            for i in range(600):
                t1 = time.perf_counter()
                yc = st.step(chunks[20 + i])
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
                ys.append(yc.clone())                   # outside the latency bracket: `step` hands back a view it overwrites
            lat.sort()
            mean = sum(lat) / len(lat)
            # parity of the streamed waveform: the 620 chunks (20 warm-up + 600 timed, carried state) against
            # the OFFLINE HIP forward of the same clip (the reference property streaming == offline), and — after
            # the cpu leg — against the reference's output of that clip
            y_stream = torch.cat(ys, dim=-1)
            y_off = net(mix8[:1], emb8[:1])[..., :y_stream.shape[-1]]
            out["_y_stream_b1"] = y_stream.cpu()
            stream_vs_offline = float((y_stream - y_off).abs().max())
            del ys, y_stream, y_off
            out["stream_b1"] = {"ms_per_chunk": mean, "p50_ms": lat[len(lat) // 2], "p99_ms": lat[int(len(lat) * 0.99)],
                                "max_ms": lat[-1], "rtf": mean / 8.0, "chunks": len(lat),
                                "parity": {"max_abs_vs_offline_hip_forward": stream_vs_offline, "chunks": 620,
                                           "tolerance_vs_offline_hip_forward": 1e-4, "ok": stream_vs_offline <= 1e-4},
                                "workload": "BASELINE configs[1]: 1 stream, 8 ms chunks (128-sample hop, 64-sample "
                                            "look-ahead), carried state, two alternating HIP graphs"}
            log(f"stream B=1: {mean:.3f} ms/chunk p99 {out['stream_b1']['p99_ms']:.3f}")
            del st
        except Exception as e:
            out["stream_b1"] = {"error": repr(e)[:200]}
        # enrollment embedder, BASELINE configs[4]
        try:
            from lookoncetohear_amd import config
            from lookoncetohear_amd.embed_net import EmbedTFGridNet
            B = 64
            enet = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
            enet.load_state_dict(config.embedder_weights(0), strict=True)
            enet = enet.to(dev)
            x = mix8.repeat(8, 1, 1).contiguous()
            ms = _time_forward(lambda: enet(x), 2, 1)
            out["_embed_b64_rows"] = enet(x)[:2].cpu()           # rows 0, 1 = utterances 0, 1: the cpu leg re-computes the first
            small = {}
            for Bs in (1, 4):                                    # one enrollment / the reference's eval batch (round 6: k_emb_inter_mv)
                xs_ = x[:Bs].contiguous()
                small[str(Bs)] = _time_forward(lambda: enet(xs_), 10, 3)
                log(f"embed B={Bs}: {small[str(Bs)]:.3f} ms")
            kern_e = embed_instrumented_step(enet, x)
            out["embed_b64"] = {"ms_per_step": ms, "clips_per_s": B / ms * 1e3, "frames_per_s": B * 1251 / ms * 1e3,
                                "workload": "BASELINE configs[4]: configs/embed.json embedder, 64 x 5 s clips (random-init "
                                            "weights; oracle front end + head pinned to reference code, trunk blocks "
                                            "restated from espnet2 — DESIGN.md §2)",
                                "ms_per_step_small_batch": small,
                                "roofline": embed_roofline(kern_e, B),
                                "kernels_ms_per_step": {k: v["total_ms"] for k, v in kern_e.items()}}
            log(f"embed B=64: {ms:.3f} ms")
            del enet, x
        except Exception as e:
            out["embed_b64"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # the eval loop's hot sequence at the headline batch (reference src/ts_hear_test.py:132-146):
        # embedder(32 enrollments) -> unsqueeze(1) -> separator(32 mixtures) -> device metric sums, one stream, both
        # workspaces resident
        try:
            from lookoncetohear_amd import config
            from lookoncetohear_amd.embed_net import EmbedTFGridNet
            from lookoncetohear_amd.metrics import metric_sums_device
            B = 32
            enet = EmbedTFGridNet(**config.EMBED_PARAMS).eval()
            enet.load_state_dict(config.embedder_weights(0), strict=True)
            enet = enet.to(dev)
            mix = mix8.repeat(4, 1, 1).contiguous()
            enr = mix8.flip(0).repeat(4, 1, 1).contiguous()          # stand-in enrollment recordings (same shape / level)
            tgt = (0.5 * mix).contiguous()
            egt = emb8.repeat(4, 1, 1).contiguous()

            def chain():
                e = enet(enr).unsqueeze(1)
                y = net(mix, e)
                return metric_sums_device(y, mix, tgt, e[:, 0], egt[:, 0])[0]

            ms = _time_forward(chain, 3, 1)
            ms_e = _time_forward(lambda: enet(enr), 3, 1)
            out["e2e_b32"] = {"ms_per_step": ms, "clips_per_s": B / ms * 1e3, "embedder_ms": ms_e, "separator_and_metrics_ms": ms - ms_e,
                              "embedder_share": ms_e / ms,
                              "workload": "enroll -> embedding -> separate -> metric sums, 32 x (5 s enrollment + 5 s mixture), one "
                                          "stream (reference src/ts_hear_test.py:132-146)"}
            log(f"e2e B=32: {ms:.3f} ms (embedder {ms_e:.3f})")
            # the same loop at the reference's own eval batch of 4 (src/ts_hear_test.py:121): both halves latency-bound there — the
            # embedder's inter axis on one workgroup per (sequence, direction), the separator in three time windows (round 6)
            mix4, enr4, tgt4, egt4 = mix[:4].contiguous(), enr[:4].contiguous(), tgt[:4].contiguous(), egt[:4].contiguous()

            def chain4():
                e = enet(enr4).unsqueeze(1)
                y = net(mix4, e)
                return metric_sums_device(y, mix4, tgt4, e[:, 0], egt4[:, 0])[0]

            ms4 = _time_forward(chain4, 10, 3)
            ms4_e = _time_forward(lambda: enet(enr4), 10, 3)
            out["e2e_b4"] = {"ms_per_step": ms4, "clips_per_s": 4 / ms4 * 1e3, "embedder_ms": ms4_e, "separator_and_metrics_ms": ms4 - ms4_e,
                             "workload": "enroll -> embedding -> separate -> metric sums at the reference's eval batch: 4 x (5 s enrollment + "
                                         "5 s mixture), one caller stream"}
            log(f"e2e B=4: {ms4:.3f} ms (embedder {ms4_e:.3f})")
            del enet, mix, enr, tgt, egt
        except Exception as e:
            out["e2e_b32"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # streaming as a product number: B independent streams in lock-step on one GPU; real-time streams
        # per MI355X at RTF <= 0.5 = B x floor-free (4 ms / p99 chunk latency)
        table = {}
        for B in (1, 8, 32, 64):
            try:
                st = net.make_streamer(B, dev, use_graph=True)
                st.set_embedding(emb8.repeat((B + 7) // 8, 1, 1)[:B, 0].contiguous())
                mixp = torch.nn.functional.pad(mix8.repeat((B + 7) // 8, 1, 1)[:B], (0, 64)).contiguous()
                chunks = [mixp[:, :, i * 128:i * 128 + 192].contiguous() for i in range(220)]
                for i in range(20):
                    st.step(chunks[i])
                torch.cuda.synchronize()
                lat = []
                for i in range(200):
                    t1 = time.perf_counter()
                    st.step(chunks[20 + i])
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t1) * 1e3)
                lat.sort()
                p50, p99 = lat[len(lat) // 2], lat[int(len(lat) * 0.99)]
                table[str(B)] = {"ms_per_chunk_p50": p50, "p99_ms": p99, "rtf_p99": p99 / 8.0,
                                 "realtime_streams_at_rtf_0.5": int(B * 4.0 / p99) if p99 > 0 else None}
                log(f"stream B={B}: p50 {p50:.3f} p99 {p99:.3f} ms")
                del st, chunks, mixp
            except Exception as e:
                table[str(B)] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
        out["stream_table"] = {"by_batch": table,
                               "note": "B streams advanced together, one graph replay per 8 ms chunk incl. the host sync a real-time "
                                       "consumer needs; realtime_streams = B x (4 ms / p99): time-multiplexed groups of B at RTF 0.5"}
        # time-axis windows of ONE batch on K streams (Net.time_chunks): same-process A/B over K, each
        # forward held bit for bit against the whole-clip one.  LAST leg on purpose: its window streams stay in the process, and
        # more streams than hardware queues (four) change how the legs above would overlap their own streams
        try:
            B = 32
            mix = mix8.repeat(4, 1, 1).contiguous()
            emb = emb8.repeat(4, 1, 1).contiguous()
            keep = net.time_chunks
            tab = {}
            net.time_chunks = 1
            y1 = net(mix, emb).clone()
            for K in (1, 2, 3, 4, 1):
                net.time_chunks = K
                same = bool(torch.equal(net(mix, emb), y1))
                ms = _time_forward(lambda: net(mix, emb), 10, 3)
                tab.setdefault(str(K), []).append(ms)
                log(f"offline B=32 time_chunks={K}: {ms:.3f} ms, bit-identical to the whole clip: {same}")
                tab[f"{K}_bit_identical"] = same
            net.time_chunks = keep
            out["offline_b32_time_chunks"] = {"ms_per_step_by_chunks": {k: v for k, v in tab.items() if not k.endswith("identical")},
                                              "bit_identical_to_whole_clip": {k[:-14]: v for k, v in tab.items() if k.endswith("identical")},
                                              "default_time_chunks": keep,
                                              "workload": "the headline batch (ONE batch in flight) with the time axis cut into K "
                                                          "windows on K HIP streams; forward only (no metric sums), 10 steps each, "
                                                          "K = 1 measured first and last"}
            del y1, mix, emb
        except Exception as e:
            out["offline_b32_time_chunks"] = {"error": repr(e)[:200]}
        net._ws.clear()
        torch.cuda.empty_cache()
    return out


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher around it: replace this process with
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same
    arguments>` — the command the contract names, one rank per GPU; the ranks then take the `WORLD_SIZE` branch of `main`.
    Under an external torchrun (RANK / WORLD_SIZE set) this is never reached."""
    import socket
    if "--dry-run-cpu" not in sys.argv and torch.cuda.device_count() < n:
        sys.exit(f"bench.py --gpus {n}: only {torch.cuda.device_count()} GPU(s) visible on this node")
    with socket.socket() as s_:                                     # a free rendezvous port on the loopback interface
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL (task environment)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: exec {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU (BASELINE configs[2] = 32)")
    ap.add_argument("--embed-batch", type=int, default=64, help="--mode embed: enrollments per GPU (BASELINE configs[4] = 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the 2 s package-power leg (N = 1 only)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="gloo / host-tensor dry run of the N>1 plumbing (no GPU, no model)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` measurements (offline B=1 / B=256, streaming, embedder, exact-fp32 contrast)")
    ap.add_argument("--mode", default="offline", choices=["offline", "stream", "embed", "render"],
                    help="offline = BASELINE configs[2] (default, the headline line); stream = configs[1]: 8 ms chunks, "
                         "carried state, HIP-graph replay per chunk (a step = one chunk)")
    ap.add_argument("--gemm", default=None, choices=["f32rec", "f16x3"], help="override Net.gemm_mode (A/B runs)")
    ap.add_argument("--no-gpu-library-baseline", action="store_true",
                    help="skip the reference-ATen-sequence-on-this-GPU context leg (PyTorch-ROCm libraries)")
    ap.add_argument("--rir-len", type=int, default=256, help="--mode render: taps per impulse response")
    ap.add_argument("--time-chunks", type=int, default=None,
                    help="override Net.time_chunks: windows of the time axis run on as many HIP streams (A/B runs; 1 = off)")
    ap.add_argument("--tune", default="", help="comma list key=value for lh_set_tuning (A/B runs), e.g. 0=2,1=1")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)                               # plain `python bench.py --gpus N`: start the N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    assert torch.cuda.is_available(), "bench.py measures the MI355X path; no GPU visible"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)     # RCCL over xGMI

    from lookoncetohear_amd import synth, _cabi
    for kv in filter(None, args.tune.split(",")):                   # A/B switches of the library (every mode)
        k, v = kv.split("=")
        assert _cabi.load().raw("lh_set_tuning")(int(k), int(v)) == 0, f"lh_set_tuning({k}, {v}) refused"
    if args.mode == "embed":
        _cabi.load()
        return bench_embed(args, dev, rank, world, dist)
    if args.mode == "render":
        _cabi.load()
        return bench_render(args, dev, rank, world, dist)
    from lookoncetohear_amd.net import Net
    from lookoncetohear_amd.metrics import metric_sums_device
    from lookoncetohear_amd import config
    _cabi.load()                                                    # fail loudly if the HIP extension is missing

    net = Net(**config.TSH_PARAMS).eval()
    net.load_state_dict(config.separator_weights(0), strict=True)   # random-init weights of the tsh.json arch
    net = net.to(dev)
    if args.gemm:
        net.gemm_mode = args.gemm
    if args.time_chunks is not None:
        net.time_chunks = args.time_chunks

    if args.mode == "stream":
        return bench_stream(args, net, dev, rank, world)

    B = args.batch
    # utterance sharding: rank r owns utterances r*B .. r*B+B-1 (seeded by index -> rank-count invariant union);
    # 8 distinct utterances are synthesised per rank and tiled to B to keep set-up time bounded
    uniq = min(B, 8)
    d = synth.batch([rank * B + i for i in range(uniq)], 80000)
    rep = (B + uniq - 1) // uniq
    mix = d["mixture"].repeat(rep, 1, 1)[:B].contiguous().to(dev)
    tgt = d["target"].repeat(rep, 1, 1)[:B].contiguous().to(dev)
    emb = d["embedding_gt"].repeat(rep, 1, 1)[:B].contiguous().to(dev)

    def step():
        y = net(mix, emb)
        sums, _ = metric_sums_device(y, mix, tgt, emb[:, 0], emb[:, 0])   # [sum si_snr_i, sum out_sisnr, sum cos, n] fp64
        if dist is not None:
            dist.all_reduce(sums)                                   # the path's only exchange step (32 B)
        return y, sums

    log = lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True)
    log(f"inputs resident: {B} clips on {dev}")
    def durations(prof):
        per = {}
        for name, e0, e1 in prof:
            per.setdefault(name, []).append(e0.elapsed_time(e1))
        return {k: dict(launches=len(v), total_ms=sum(v), avg_ms=sum(v) / len(v)) for k, v in per.items()}

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        # one fully instrumented step (HIP events around every C-ABI call): the per-call breakdown, and which call
        # dominates.  The timed region then brackets only the dominant call's launches with events, so the 27 calls of a
        # step are not separated by 54 event records.
        net._prof = []
        step()
        torch.cuda.synchronize()
        breakdown = durations(net._prof)
        dom = max(breakdown, key=lambda k: breakdown[k]["total_ms"])
        log(f"warm-up done; dominant call {dom}")
        # Sampled: every 4th call of the dominant entry point is bracketed (3 calls per step, so the samples rotate through
        # the three blocks).  A HIP event record is not free on the stream: the dispatch timeline of a step showed a 5.6-6.3 us
        # idle gap at each of the six records of a fully bracketed step (profiles/r05o_b32_step_timeline.txt: 35 us = 0.5 % of
        # the step was the measurement itself); sampled, the bracket costs ~9 us per step.
        net._prof, net._prof_only, net._prof_stride, net._prof_count = [], {dom}, int(os.environ.get("LOOKONCE_BENCH_EVENT_STRIDE", "4")), {}
        elapsed, local_elapsed, (y, sums) = timed_region(step, args.steps, dist, dev, torch.cuda.synchronize)
        prof, net._prof, net._prof_only, net._prof_stride = net._prof, None, None, 1
        roof_samples = len(prof)
    log(f"timed region: {local_elapsed * 1e3 / args.steps:.3f} ms/step (max over ranks {elapsed * 1e3 / args.steps:.3f})")
    # the forwards ran asynchronously (deferred range check, net.py): one look at this Net's flag word for the whole region
    range_flag_raised = bool(net.range_status(dev)) if net.range_check else None
    assert not range_flag_raised, "LH_ERR_RANGE: a timed forward stored zeros in place of non-finite samples"
    # N > 1: the exchange step by itself — 100 all-reduces of the 32-byte metric vector on RCCL, HIP events on this rank
    allreduce_us, ranks_seen = None, world
    if dist is not None:
        ranks_seen = dist.get_world_size()
        buf = sums.clone()
        for _ in range(10):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        allreduce_us = e0.elapsed_time(e1) * 10.0
    power = None
    if world == 1 and not args.no_power:
        with torch.no_grad():
            power = power_leg(step)
        log(f"power leg: {power}")

    # HIP-event durations (this rank): the dominant call live over the timed region, the others from the instrumented
    # warm-up step
    kern = dict(breakdown)
    kern.update(durations(prof))
    gpu_ms = sum(v["avg_ms"] * breakdown[k]["launches"] for k, v in kern.items())

    if rank == 0:
        total_clips = world * B * args.steps
        value = total_clips * FRAMES_PER_CLIP / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        w = KERNEL_WORK[dom]
        traffic, traffic_src, tj_dom = None, None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # HBM bytes per launch from the PMC passes
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            if tj.get("batch_per_gpu") == B and dom in tj.get("kernels", {}):
                tj_dom = tj["kernels"][dom]
                traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]    # per kernel launch (rocprofv3 dispatch)
                traffic_src = ("profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "bench, committed with the kernels; source profile: %s) — not re-measured in this run"
                               % tj.get("source", "see profiles/README.md"))
        if w["bound"] == "mfma":
            ach = w["flops"] * B / (kern[dom]["avg_ms"] * 1e-3) / 1e12
            peak = PEAK_F16_MFMA_TFLOPS if net.gemm_mode == "f16x3" else PEAK_FP32_MFMA_TFLOPS
            roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                        traffic=traffic,
                        note=("algorithmic fp32-equivalent FLOPs; f16x3 mode executes 3x as many fp16 MFMA FLOPs, "
                              "peak = dense fp16 MFMA") if net.gemm_mode == "f16x3" else "exact fp32 MFMA")
        else:
            ach = w["bytes"] * B / (kern[dom]["avg_ms"] * 1e-3) / 1e9
            roof = dict(kernel=dom, bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=ach / PEAK_HBM_GBS, traffic=traffic)
        lpc = LAUNCHES_PER_CALL.get(dom, 1)
        roof["traffic_source"] = traffic_src
        roof["kernel_function"] = KERNEL_NAME.get(dom, dom)
        roof["launches_per_call"] = lpc            # achieved = work of one call / duration of one call (= per launch too)
        roof["avg_call_ms"] = kern[dom]["avg_ms"]
        roof["event_samples"] = roof_samples       # calls of the dominant entry point bracketed with HIP events in the timed region
        roof["avg_launch_ms"] = kern[dom]["avg_ms"] / lpc
        # the same call in the fully instrumented warm-up step (event pairs around every call, as rocprofv3's tracer
        # also separates the dispatches): the figure to hold against profiles/*kernel_stats*.csv
        roof["avg_launch_ms_instrumented_step"] = breakdown[dom]["avg_ms"] / lpc
        roof["share_of_gpu_time"] = kern[dom]["avg_ms"] * breakdown[dom]["launches"] / gpu_ms
        # the same kernel against the other roofline, and against what the chip sustains under its 1400 W package limit
        roof["frac_hbm"] = w["bytes"] * B / (kern[dom]["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
        if w["bound"] == "mfma" and net.gemm_mode == "f16x3":
            roof["executed_fp16_tflops"] = 3.0 * roof["achieved"]
            roof["frac_of_measured_mfma_rate_at_power_cap"] = 3.0 * roof["achieved"] / MEASURED_MFMA16_TFLOPS_AT_CAP
        if tj_dom is not None:
            for k_ in ("mfma_busy", "valu_busy"):
                if k_ in tj_dom:
                    roof[k_] = tj_dom[k_]
            if "ratio_to_survey_8d_per_call" in tj_dom:     # the contract's byte count for the stage (2 A), next to the kernel's own
                roof["traffic_ratio_to_survey_8d_stage_bytes"] = tj_dom["ratio_to_survey_8d_per_call"]
                roof["traffic_note"] = ("traffic = counter bytes per launch; two launches per call move 5 A by construction (forward: "
                                        "read x, write out; reverse: read x, read + rewrite out) against SURVEY §8(d)'s 2 A for the stage")
            if "rocprof_avg_launch_us" in tj_dom:
                # the committed rocprofv3 --kernel-trace --stats summary times the same launch with every dispatch separated by
                # the tracer (and at its own clocks): ~10 % longer than the HIP-event figure of back-to-back launches above.
                # Both are stated; `frac` is the HIP-event one.
                rp = tj_dom["rocprof_avg_launch_us"] * 1e-3
                roof["rocprof_avg_launch_ms"] = rp
                roof["frac_at_rocprof_duration"] = roof["frac"] * roof["avg_launch_ms"] / rp
                roof["duration_note"] = ("avg_launch_ms: HIP events on the launch stream, launches back to back (this run); "
                                         "rocprof_avg_launch_ms: profiles/*kernel_stats.csv of the committed profile, dispatches separated by the tracer")
        roof["limited_by"] = limited_by(power, roof)
        out = {
            "metric": "frames_per_sec (5 s 16 kHz binaural clips, 625 STFT frames each, offline forward)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 via f16x3 split (3x fp16 MFMA per product, fp32 accumulate; ~22-bit operands, un-normalised rows "
                      "scaled by a power of two before the split: any finite fp32 range)"
                      if net.gemm_mode == "f16x3" else "f32 (exact fp32 MFMA recurrences, split-precision frame kernels)"),
            "data": "synthetic",
            "rtf": elapsed / (total_clips * CLIP_SECONDS),
            "config": {"workload": f"BASELINE configs[2]: {B} x 5 s 16 kHz binaural clips per GPU, offline forward "
                                   f"(configs/tsh.json separator, random-init weights)",
                       "batch_per_gpu": B, "global_batch": world * B, "parallelism": f"utterance-dp{world}",
                       "gemm_mode": net.gemm_mode, "tune": args.tune,
                       "time_chunks": net._n_time_chunks(B, FRAMES_PER_CLIP, 1 if net.gemm_mode == "f16x3" else 0),
                       "batches_in_flight": 1},
            "whole_path": {"algorithmic_tflops": FLOPS_PER_CLIP * total_clips / elapsed / 1e12,
                           "algorithmic_hbm_gbs": (BYTES_PER_CLIP * B + WEIGHT_BYTES) * world * args.steps / elapsed / 1e9,
                           "frac_hbm_peak": (BYTES_PER_CLIP * B + WEIGHT_BYTES) * world * args.steps / elapsed / 1e9 / (PEAK_HBM_GBS * world)},
            "roofline": roof,
            "power": power,
            "n_ranks_seen": ranks_seen, "allreduce_32B_us": allreduce_us,
            "range_check": {"mode": "deferred" if net.range_check is True else net.range_check, "flag_raised": range_flag_raised},
            # ms per step of each C-ABI call: the dominant one live over the timed region, the rest from the instrumented
            # warm-up step
            "kernels_ms_per_step": {k: v["avg_ms"] * breakdown[k]["launches"]
                                    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["avg_ms"] * breakdown[kv[0]]["launches"])},
            "metric_sums": [float(v) for v in sums.tolist()],
        }
        emb_rows, kept = None, {}
        if not args.no_secondary and world == 1 and args.batch == 32 and not args.gemm:
            out["secondary"] = secondary_measurements(net, dev, mix[:8], emb[:8])
            emb_rows = out["secondary"].pop("_embed_b64_rows", None)
            kept = {k: out["secondary"].pop(k) for k in [k for k in out["secondary"] if k.startswith("_y_")]}
        out["parity"] = None
        if not args.no_cpu_baseline and world == 1:
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                dump = os.path.join(tmp, "cpu_leg.npz")
                out["cpu_baseline"] = cpu_baseline_subprocess(dump=dump)
                emb_cpu = out["cpu_baseline"].pop("embed_cpu_baseline", None)
                if os.path.exists(dump):
                    # parity inside the driver-run line: HIP vs the reference outputs the cpu leg just produced
                    try:
                        out["parity"], ref = parity_object(net, dev, dump)
                        log(f"parity: {out['parity']}")
                        if "secondary" in out:
                            secondary_parity(out["secondary"], kept, ref)
                            log("secondary parity: " + json.dumps({k: v.get("parity") for k, v in out["secondary"].items()
                                                                   if isinstance(v, dict) and "parity" in v}))
                        sec = out.get("secondary", {}).get("embed_b64")
                        if isinstance(sec, dict) and emb_rows is not None and "emb" in ref.files:
                            er = torch.from_numpy(ref["emb"]).double()
                            eh = emb_rows[:er.shape[0]].double()
                            cos = torch.nn.functional.cosine_similarity(eh, er, dim=-1)
                            sec["embedding_cos_min"] = float(cos.min())
                            sec["embedding_max_abs"] = float((eh - er).abs().max())
                            sec["parity_clips"] = int(er.shape[0])
                            sec["parity_against"] = ("oracle/embedder_oracle.py fp32 on the full-length enrollment(s) of the first row(s) "
                                                     "(one 5 s clip costs the CPU ~10^4 x the GPU's time; trunk restated from espnet2: "
                                                     "parity unpinned, DESIGN.md §2)")
                            sec["cpu_baseline"] = emb_cpu
                    except Exception as e:
                        out["parity"] = {"error": repr(e)[:300]}
        else:
            out["cpu_baseline"] = None
        # context only: the reference's op sequence on this GPU's libraries (never `vs_baseline`: BASELINE.md holds no
        # published number for this metric, so that field stays null by contract)
        out["gpu_library_baseline"] = None
        if world == 1 and not args.no_gpu_library_baseline and not args.no_cpu_baseline:
            net._ws.clear()
            torch.cuda.empty_cache()
            g = gpu_library_baseline_subprocess()
            out["gpu_library_baseline"] = g
            if g.get("value"):
                out["vs_gpu_library_baseline"] = value / g["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
