#!/usr/bin/env python
"""Headline benchmark of the MAEST hot path on MI355X (BASELINE.json: clips/sec, 10 s @ 16 kHz,
96-mel, MAEST-10s fwd+bwd).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training step of BASELINE config 3 on ONE batch of synthetic input already
resident in HBM: mixup draw -> fused mixup/patchout/patch-embed -> 12-block ViT forward -> BCE ->
hand-written backward -> (N > 1: RCCL gradient all-reduce over xGMI, overlapped) -> AdamW.
Per-GPU batch is fixed (weak scaling); `value` is the whole-job clips/s.  Rank 0 prints ONE JSON line
that also carries
  "roofline":     the dominant kernel (bf16 MFMA GEMM) priced against the 2.5 PFLOP/s dense bf16 peak,
                  timed live with HIP events on the launch stream inside the timed region;
  "cpu_baseline": the oracle (CPU restatement of the reference, oracle/maest_oracle.py -- proven
                  bit-identical to the imported reference) timed on this box's host cores on a bounded
                  sample of the same workload.  A reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import gc
import time

# before the first HIP call: see maest_amd/__init__.py.  (--ranks-share-gpu, the path test that puts all N processes on ONE device, takes TWO
# queues per process: the device's hardware queue slots are oversubscribed by eight processes, the scheduler time-slices them by saving and
# restoring waves, and with 32 - 64 queues of kernels that own whole CUs a rank died with a GPU fault (illegal instruction / memory fault in an
# unrelated copy kernel) in about one run of eight -- profiles/r06_hw_queues.txt.  Not a configuration anything but that test runs in.)
if "--ranks-share-gpu" in sys.argv:
    os.environ["GPU_MAX_HW_QUEUES"] = "2"
else:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def flops_per_clip_fwd(N, Tk, C=400):
    per_block = 2 * N * 768 * (2304 + 768 + 3072 + 3072) + 4 * N * N * 768
    return 12 * per_block + 2 * (9 * Tk) * 256 * 768 + 2 * 768 * C


def flops_per_clip_fwd_not_executed(N, attn_rows):
    """Forward FLOPs of the algorithmic count above that the engine does not execute: the last block runs its attention
    queries, proj and MLP only for the two tokens the head reads (maest.py: _Engine.head_tail; outputs and gradients are
    those of the full evaluation).  attn_rows = query rows the attention kernel still computes per clip."""
    return 2 * (N - 2) * 768 * (768 + 3072 + 3072) + 4 * (N - attn_rows) * N * 768


PMC_TRAFFIC_FILE = "profiles/r03c_pmc_traffic.json"


def pmc_traffic():
    """FALLBACK of measure_pmc_traffic(): (HBM bytes per launch of the dominant kernel, where that number comes from) read
    from the committed summary of separate rocprofv3 --pmc passes over THIS command (scratch/profile_round.sh), i.e. a
    constant from an earlier run of the same binary, labelled as such in the JSON line (`traffic_source`).  (None, reason)
    when the file is absent."""
    for name in (PMC_TRAFFIC_FILE, "profiles/r02c_pmc_traffic.json"):
        try:
            with open(os.path.join(REPO, name)) as f:
                return json.load(f)["traffic_bytes_per_launch"], (
                    f"{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed "
                    "(2 x FETCH_SIZE + WRITE_SIZE per launch); not measured by this run")
        except Exception:
            continue
    return None, "no committed PMC summary found"


def measure_pmc_traffic(timeout_s=150):
    """HBM bytes per GEMM call of the dominant kernel, MEASURED by this run: two rocprofv3 passes (--kernel-trace --pmc
    FETCH_SIZE, then WRITE_SIZE: counters in their own passes, no other trace domain) over a short child run of this same
    file (2 steps + 1 warm-up, kernels serialized, no side cases), on this GPU.  Per call = sum over every gemm_nt256o /
    gemm_nt256w launch (256-row tiles and the 128-row-tile launch behind some of them) / number of 256-row-tile launches;
    traffic = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half of the wide coalesced
    reads on gfx950; profiles/r03c_pmc_traffic.json holds the same passes with their calibration on known byte counts:
    x2.00 / x1.00).  Returns (bytes, source, parts) or (None, reason, None): any failure falls back to the committed figure."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH", None
    parts = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="maest_pmc_", dir="/tmp") as d:
                cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
                       sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                       "--no-kernel-timing", "--serial-kernels", "--no-side-cases"]
                env = dict(os.environ, TMPDIR="/tmp", MAEST_BENCH_PMC="0")
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
                if r.returncode != 0:
                    return None, f"rocprofv3 --pmc {counter} pass exited with {r.returncode}", None
                total, calls = 0.0, 0
                for f in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if ("gemm_nt256w_kernel" not in row["Kernel_Name"] and "gemm_nt256o_kernel" not in row["Kernel_Name"]) \
                                or row["Counter_Name"] != counter:
                            continue
                        total += float(row["Counter_Value"])
                        name = row["Kernel_Name"].split("(")[0].replace(" ", "")
                        # the 256-row-tile kernels (bf16: gemm_nt256o; otherwise gemm_nt256w<.., 4>): one launch per GEMM call
                        if "gemm_nt256o_kernel" in name or name.endswith(",4>"):
                            calls += 1
                if calls == 0:
                    return None, f"no gemm_nt256o / gemm_nt256w launches in the --pmc {counter} pass", None
                parts[counter] = (total * 1024.0 / calls, calls)
    except Exception as e:      # timeout, parse error, ...
        return None, f"live PMC pass failed: {type(e).__name__}: {e}", None
    fetch, write = parts["FETCH_SIZE"][0], parts["WRITE_SIZE"][0]
    src = ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) over a child "
           "`bench.py --steps 2 --warmup 1 --serial-kernels` of the same binary on this GPU; 2 x FETCH_SIZE + WRITE_SIZE per "
           f"GEMM call ({parts['FETCH_SIZE'][1]} calls profiled)")
    return 2.0 * fetch + write, src, {"fetch_size_raw_per_call": round(fetch), "write_size_per_call": round(write)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU
    (what Lightning does for the reference, ex_maest.py:49,57).  Rank 0 of the child job prints the JSON line."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def time_mel_kernel(dev, clips, samples, reps=10):
    """logmel_kernel (csrc/mel.hip) alone, HIP events on its launch stream: algorithmic bytes = fp32 waveform in +
    fp32 [96, T] out (0.88 MB per 10 s clip) over the average launch time, against the 8 TB/s HBM peak."""
    from maest_amd.melspectrogram import MelSpectrogram
    mel = MelSpectrogram()
    w = torch.rand((clips, samples), device=dev) * 2 - 1
    for _ in range(2):
        out = mel(w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        out = mel(w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = w.numel() * 4 + out.numel() * 4
    return {"kernel": "logmel_kernel (framing + Hann + 512-pt FFT + |.|^2 + mel + logC + z-norm, one pass)",
            "bound": "hbm", "clips": clips, "samples": samples, "frames": int(out.shape[-1]),
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes": nbytes,
            "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4), "clips_per_s": round(clips / (ms * 1e-3), 0)}


def _median_times(fn, steps):
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        fn(it)
        if it > 0:
            times.append(time.perf_counter() - t0)
    return float(np.median(times))


def _best_thread_count(fn, steps):
    """The oracle pass `fn` under torch.set_num_threads(c) for c in {8, 16, 32, 64 (, all host cores if fewer)}: one warm-up pass, then ONE timed pass
    per count (a batch of 8 clips oversubscribes 128 threads: round 5's all-cores figure was below an 8-core run of the reference
    itself), then the best count again until it has `steps` passes.  Returns (median seconds per pass at the best count, best count,
    {count: seconds} of the sweep).  The thread count is left at the best one (the CPU baseline is the last thing the bench does)."""
    ncpu = os.cpu_count() or 1
    # (all host threads is part of the sweep only up to 64: on the 256-thread hosts of this pool one batch-8 step took 165 s that way against 1.0 s at
    # 16 threads -- profiles/r06a_bench_default_line.json has that sweep -- and the default line has to finish within minutes)
    cands = sorted({c for c in (8, 16, 32, 64, ncpu if ncpu <= 64 else 64) if c <= ncpu})
    torch.set_num_threads(cands[0])
    fn(0)                                   # warm-up: allocations, thread pool
    sweep = {}
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        fn(1)
        sweep[c] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]]
    while len(times) < steps:
        t0 = time.perf_counter()
        fn(len(times) + 1)
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), best, {str(c): round(t, 2) for c, t in sweep.items()}


def cpu_baseline_infer(batch, T, steps=3):
    """Oracle eval forward on the host cores (B = 8, N = 560)."""
    from oracle import maest_oracle as O
    img_t = 625 if T <= 640 else (T // 5) * 5
    sd = O.make_state_dict(img_t, seed=1234)
    rng = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(rng.standard_normal((batch, 96, T), dtype=np.float32))

    def one(it):
        with torch.no_grad():
            O.forward(x, sd, (96, img_t), melspectrogram_input=True)
    t, best, sweep = _best_thread_count(one, steps)
    return {"value": round(batch / t, 3), "unit": "clips/s", "cores": best, "host_cores": os.cpu_count(), "kind": "port",
            "thread_sweep_s_per_pass": sweep,
            "sample": f"oracle eval forward (fp32) batch={batch} T={T}; best of a thread sweep {sorted(int(k) for k in sweep)} (one pass each after 1 warm-up), "
                      f"then the median of {steps} passes at {best} threads: {t:.2f} s/pass"}


def cpu_baseline(batch, T, patchout, steps=3, teacher_student=False, waveform=False):
    """Oracle training step (fwd + bwd + AdamW) on the host cores: the reference's algorithm.  teacher_student:
    C = 519, separated heads, (BCE + BCE) / 2 (models/module.py:280-316); waveform: the log-mel front end is part
    of the step (BASELINE configs[4])."""
    from oracle import maest_oracle as O
    torch.manual_seed(0)
    img_t = 625 if T <= 640 else (T // 5) * 5
    C = 519 if teacher_student else 400
    sd = {k: v.clone().requires_grad_(True) for k, v in O.make_state_dict(img_t, n_classes=C, seed=1234).items()}
    params = [v for k, v in sd.items() if teacher_student or not k.startswith("head_dist")]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=1e-4)
    rng = np.random.Generator(np.random.PCG64(3))
    if waveform:
        wav = torch.from_numpy(rng.random((batch, (T - 1) * 256), dtype=np.float32) * 2 - 1)
    else:
        x = torch.from_numpy(rng.standard_normal((batch, 1, 96, T), dtype=np.float32))
    y = torch.from_numpy((rng.random((batch, C)) < 2.5 / C).astype(np.float32))
    yt = torch.from_numpy((rng.random((batch, C)) < 2.5 / C).astype(np.float32)) if teacher_student else None
    Tp = (T - 16) // 10 + 1

    def one(it):
        perm = torch.randperm(batch)
        lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, batch).astype(np.float32), 1 - b))
        keep = torch.randperm(Tp)[: Tp - patchout].sort().values.tolist()
        xin = O.logmel(wav).unsqueeze(1) if waveform else x
        loss = O.training_loss(xin, y, sd, perm, lam, toffset=0, t_keep=keep, y_teacher=yt)[0]
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    t, best, sweep = _best_thread_count(one, steps)
    what = ("oracle teacher-student training step (log-mel + fwd + bwd + AdamW, fp32)" if teacher_student
            else "oracle training step (fwd+bwd+AdamW, fp32)")
    return {"value": round(batch / t, 3), "unit": "clips/s", "cores": best, "host_cores": os.cpu_count(), "kind": "port",
            "thread_sweep_s_per_step": sweep,
            "sample": f"{what} batch={batch} T={T} patchout={patchout}; best of a thread sweep {sorted(int(k) for k in sweep)} (one step each after 1 warm-up), "
                      f"then the median of {steps} steps at {best} threads: {t:.2f} s/step"}


def build_case(args, dev, rank, world, mode, T, B, patchout, reducer_kw=None):
    """Model + resident synthetic batch + the step closure of one bench configuration."""
    from maest_amd import get_maest
    from maest_amd.dist import GradReducer, broadcast_parameters
    from maest_amd.module import Module, TeacherStudentModule
    ts = mode == "ts"
    train = mode in ("train", "ts")
    C = 519 if ts else 400
    img_t = (T // 5) * 5 if T > 640 else 625      # time table: 62 columns for 10 s, 187 for the 30 s configs
    arch = "discogs-maest-30s-pw-73e-ts" if ts else ("passt_s_swa_p16_128_ap476" if train else "discogs-maest-10s-pw-129e")
    net = get_maest(arch, pretrained=False, input_t=img_t, n_classes=C, s_patchout_t=patchout,
                    distilled_type="separated" if ts else "mean", precision=args.precision).to(dev)
    broadcast_parameters(net)
    if args.complete_last_block:
        net._engine.head_tail = False
    if args.serial_kernels:
        net._engine.overlap_wgrad = False
        net.eval_streams = 1
    if args.no_fold_delta:
        net._engine.fold_delta = False
    mod = (TeacherStudentModule if ts else Module)(net=net, mixup_alpha=0.3)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    if ts:   # 30 s of synthetic 16 kHz audio per clip; the log-mel front end runs inside every step
        x = torch.rand((B, (T - 1) * 256), generator=gen, device=dev) * 2 - 1
    else:
        x = torch.randn((B, 1, 96, T), generator=gen, device=dev)               # synthetic z-normed log-mel
    y = (torch.rand((B, C), generator=gen, device=dev) < 2.5 / C).float()       # ~2.5 labels per clip
    y_teacher = (torch.rand((B, C), generator=gen, device=dev) < 2.5 / C).float() if ts else None
    if train:
        net.train()
        if args.hip_graph:
            net.enable_hip_graph()
        opt = mod.get_optimizer(net.parameters())
        reducer = None
        if world > 1 or args.force_collective:
            skip = () if ts else ("head_dist.weight", "head_dist.bias")
            reducer = GradReducer(net.named_parameters(), skip=skip, force_collective=args.force_collective)
            reducer.timing = os.environ.get("MAEST_DP_BUCKET_TIMING", "1") != "0"      # two event records per bucket: launch -> complete of every all-reduce, printed as `dp_buckets`
            net._grad_sink = reducer
        batch = (x, None, y, y_teacher) if ts else (x, None, y)

        # precision="fp16": the loop the reference's trainer runs under 16-mixed (ex_maest.py:51) -- torch.amp.GradScaler around the optimizer:
        # scaled loss, gradients unscaled and checked for inf / nan (one host read per step), the step skipped and the scale halved on overflow
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if args.precision == "fp16" else None

        def step(xin=None):
            if reducer is not None:
                reducer.reset()
            loss = mod.training_step(batch if xin is None else (xin,) + tuple(batch[1:]), 0)
            (loss if scaler is None else scaler.scale(loss)).backward()
            if reducer is not None:
                reducer.finish()
            if scaler is None:
                opt.step()
            else:
                scaler.step(opt)
                scaler.update()
            if reducer is None:      # (with a reducer the gradients are views of its flat buffer, zeroed by reset())
                opt.zero_grad(set_to_none=True)
            return loss
    else:
        net.eval()
        if args.hip_graph:
            net.enable_hip_graph()

        def step():
            with torch.no_grad():
                return net(x)[0]
    Tp = (T - 16) // 10 + 1
    Tk = Tp - patchout
    return dict(net=net, step=step, mode=mode, ts=ts, train=train, T=T, B=B, patchout=patchout, C=C, arch=arch,
                Tk=Tk, N=2 + 9 * Tk, x=x)


def deviation_vs_fp32(case, precision, clips=4):
    """How far the timed numeric mode is from the exact-fp32 mode of the SAME weights on clips of the SAME batch: relative
    error of the logits (max |a - b| / max |b|), one forward each under no_grad, with the patchout draw of a training
    configuration pinned so that both forwards see the same tokens.  (The reference computes in fp32 on the CPU; the fp32
    mode matches it to ~1e-6, tests/test_model_gpu.py G1 -- no oracle in this path.)"""
    net, x = case["net"], case["x"][:clips]
    po = None
    if case["train"] and case["patchout"] > 0:
        Tp = (case["T"] - 16) // 10 + 1
        keep = torch.sort(torch.randperm(Tp, generator=torch.Generator().manual_seed(11))[:Tp - case["patchout"]]).values
        po = (0, keep)
    prev = net.precision
    try:
        with torch.no_grad():
            net.precision = precision
            a = net(x.clone(), _patchout=po)[0].float()
            net.precision = "fp32"
            b = net(x.clone(), _patchout=po)[0].float()
    finally:
        net.precision = prev
    torch.cuda.synchronize()
    return {"logits_rel_err": float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()), "clips": int(x.shape[0]),
            "against": "precision=\"fp32\" (exact fp32 MFMAs) of the same weights on the same clips, same tokens"}


def check_ranks(net, world, dev):
    """--check-ranks: every rank's weights as (sum, sum of squares, xor of the bit patterns) gathered on all ranks and
    compared exactly -- data-parallel replicas that saw the same averaged gradients hold bit-identical weights --, and the
    bucket all-reduces the last step launched."""
    import torch.distributed as dist
    flat = torch.cat([p.detach().reshape(-1).float() for p in net.parameters()])
    bits = flat.view(torch.int32)
    # xor-fold of all bit patterns (exact, order-independent)
    n = 1 << (bits.numel() - 1).bit_length()
    pad = torch.zeros(n, dtype=torch.int32, device=dev)
    pad[:bits.numel()] = bits
    while pad.numel() > 1:
        half = pad.numel() // 2
        pad = pad[:half] ^ pad[half:]
    sig = torch.stack([flat.double().sum(), (flat.double() ** 2).sum(), pad[0].double()])
    red = getattr(net, "_grad_sink", None)
    res = {"buckets": None if red is None else len(red.buckets),
           "all_reduces_last_step": None if red is None else int(red.reduced_this_step)}
    if world > 1:
        sigs = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        res["weights_identical_across_ranks"] = bool(all(torch.equal(s, sigs[0]) for s in sigs))
        res["ranks_compared"] = world
    else:
        res["weights_identical_across_ranks"] = True
        res["ranks_compared"] = 1
    return res


def timed_steps(step, steps, warmup, world, dev):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # Host hygiene of any long-running training loop: a full (generation 2) Python collection walks every object torch / numpy / the model
    # ever created -- 120 .. 170 ms measured here, longer than the three to four steps the engine lets the host run ahead of the device
    # (maest_amd/maest.py: run_ahead).  Collect once now and park the survivors in the permanent generation; the cyclic collector keeps running
    # during the timed steps on what they allocate.
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


TIMED_KINDS = {"maest_gemm_nt", "maest_gemm_nt_small", "maest_gemm_tn", "maest_attn_fwd", "maest_attn_bwd",
               "maest_layernorm_fwd", "maest_layernorm_bwd", "maest_logmel"}


def kernel_pass(case, steps):
    """The same K steps again with HIP events around every launch of the GEMM / attention / LayerNorm kernels on the
    launch stream, kernels serialized (side-stream overlap off) so that each event pair times one kernel alone.  Kept
    out of the timed region: ~400 event records per step cost ~10 % of a step on the host."""
    from maest_amd import ops
    net = case["net"]
    prev, prev_streams = net._engine.overlap_wgrad, net.eval_streams
    net._engine.overlap_wgrad = False
    net.eval_streams = 1          # (evaluation forwards of large batches run as two half batches on two streams: MAEST._eval_forward)
    net.enable_hip_graph(False)
    with ops.KernelTimer(kinds=TIMED_KINDS) as timer:
        for _ in range(steps):
            case["step"]()
    torch.cuda.synchronize()
    net._engine.overlap_wgrad, net.eval_streams = prev, prev_streams
    return timer


def flop_counts(case, precision):
    """(algorithmic step FLOPs, FLOPs of it the engine does not execute, query rows the last block's attention computes)."""
    from maest_amd import ops
    net, N, Tk, C, B, train, ts = case["net"], case["N"], case["Tk"], case["C"], case["B"], case["train"], case["ts"]
    fwd = flops_per_clip_fwd(N, Tk, C) + (2 * 768 * C if ts else 0)
    step_flops = (3 if train else 1) * fwd * B
    tail_on = bool(getattr(net._engine, "head_tail", False))
    # (training at shapes the fused attention backward does not serve keeps the last block's attention complete)
    attn_rows = min(32, N) if (not train or ops.attn_bwd_rows_supported(
        torch.bfloat16 if precision in ("bf16", "fp16") else torch.float32, N)) else N
    skipped = (3 if train else 1) * flops_per_clip_fwd_not_executed(N, attn_rows) * B if tail_on else 0
    return step_flops, skipped, attn_rows, tail_on


def kernel_report(case, timer, steps, precision, with_traffic, live_traffic=None):
    """roofline / kernel_ms_per_step / attention_set objects from one kernel_pass."""
    out = {}
    B, N, train = case["B"], case["N"], case["train"]
    _, _, attn_rows, tail_on = flop_counts(case, precision)
    summ = timer.summary()
    g = summ.get("maest_gemm_nt")
    if g and g["ms"] > 0:
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12
        # bf16x3 spends 3 bf16 MFMAs per product: its algorithmic rate is priced against a third of the bf16 peak
        peak = {"bf16": PEAK_BF16_TFLOPS, "fp16": PEAK_BF16_TFLOPS, "bf16x3": round(PEAK_BF16_TFLOPS / 3, 1)}.get(precision, 157.3)
        from maest_amd import _lib as _L
        ow = bool(_L.kernel_forms() & _L.FORM_GEMM_NT_OW)      # (False: a build whose register audit failed keeps the eight-wave kernel)
        out["roofline"] = {"bound": "mfma", "kernel": (("maest_gemm_nt (gemm_nt256o_kernel, bf16, one wave per SIMD -- plus gemm_nt256w_kernel<bf16, 2> for the 128-row tail tiles: every token-major GEMM of the blocks; the head and the last block's head-token rows, M <= 512, run gemm_nt_kernel and are listed as maest_gemm_nt_small)"
                                                       if ow else "maest_gemm_nt (gemm_nt256w_kernel<bf16>, eight waves: this build left the one-wave-per-SIMD kernel out -- maest_kernel_forms())")
                                                      if precision in ("bf16", "fp16") else
                                                      ("maest_gemm_nt (3 bf16 MFMAs per fp32 product: all four linears of a block as ONE bf16 GEMM over 3 K on gemm_nt256o_kernel -- "
                                                       "split operand rows, MAEST_SPLIT3_A x MAEST_SPLIT3_B --, the last block's head rows on gemm_nt256w_kernel<float, X3>; "
                                                       "algorithmic flops 2 M N K)"
                                                       if precision == "bf16x3" else "maest_gemm_nt (fp32 MFMA)")),
                           "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
        if with_traffic:
            if live_traffic is not None and live_traffic[0] is not None:
                out["roofline"].update(traffic=live_traffic[0], traffic_source=live_traffic[1], traffic_parts=live_traffic[2])
            else:
                traffic, traffic_source = pmc_traffic()
                if live_traffic is not None:
                    traffic_source += f" (live measurement unavailable: {live_traffic[1]})"
                out["roofline"].update(traffic=traffic, traffic_source=traffic_source)
        out["roofline"].update(launches_per_step=g["launches"] // steps, avg_launch_ms=round(g["ms"] / g["launches"], 4),
                               ms_per_step=round(g["ms"] / steps, 3),
                               note="second pass over the same K steps, kernels serialized (side-stream overlap off) "
                                    "so that each event pair times one kernel alone")
    out["kernel_ms_per_step"] = {k: round(v["ms"] / steps, 3) for k, v in summ.items()}
    att = [summ.get("maest_attn_fwd"), summ.get("maest_attn_bwd")]
    aw = sum(a["work"] for a in att if a)
    am = sum(a["ms"] for a in att if a)
    if am > 0:
        out["attention_core_tflops"] = round(aw / (am * 1e-3) / 1e12, 1)
    # north_star: "fraction of the attention-GEMM roofline" = the 12-block attention set (QKV projection + QK^T + PV +
    # output projection), its launches picked out of the timed records by their algorithmic work (the MLP
    # GEMMs have 3072-wide shapes)
    Mtok = B * N
    set_work = {2.0 * Mtok * 2304 * 768, 2.0 * Mtok * 768 * 768, 2.0 * (2 * B) * 768 * 768}
    set_ms = sum(e0.elapsed_time(e1) for name, e0, e1, w in timer.records
                 if name in ("maest_attn_fwd", "maest_attn_bwd")
                 or (name in ("maest_gemm_nt", "maest_gemm_nt_small", "maest_gemm_tn") and w in set_work)) / steps
    set_flops = (3 if train else 1) * B * 12 * (2.0 * N * 768 * 2304 + 4.0 * N * N * 768 + 2.0 * N * 768 * 768)
    if tail_on:       # executed FLOPs: the last block's out-projection and attention queries cover the head tokens only
        set_flops -= (3 if train else 1) * B * (2.0 * (N - 2) * 768 * 768 + 4.0 * (N - attn_rows) * N * 768)
    if set_ms > 0:
        out["attention_set"] = {"what": "12 x (QKV proj + QK^T + PV + out proj)" + (", fwd+bwd" if train else ", fwd"),
                                "ms_per_step": round(set_ms, 3), "tflops": round(set_flops / set_ms / 1e9, 1),
                                "mfma_frac": round(set_flops / set_ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
    return out


def loader_case(args, dev, steps=10, B=256):
    """SURVEY 8f row 1 on the line: the on-disk mel reader (raw float16 [frames, 96] files, random offsets, one pinned staging
    buffer + one H2D copy + melfile_assemble_kernel: maest_amd/melfile.py; reference: discogs/dataset.py:69-140 + the norm of
    discogs/datamodule.py:126-152, 16 host workers per GPU there) feeding the configs[2] training step: B files of 60 s
    (3750 frames) on tmpfs, 625-frame clips.  Timed alone, and double-buffered -- a producer thread assembling batch i + 1 on
    a second stream while the step consumes batch i."""
    import queue
    import shutil
    import tempfile
    import threading
    from maest_amd.melfile import MelFileReader
    root = tempfile.mkdtemp(prefix="maest_loader_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rng = np.random.Generator(np.random.PCG64(99))
        names = []
        for i in range(B):
            (rng.standard_normal((3750, 96), dtype=np.float32) * 0.4 + 2.0).astype(np.float16).tofile(os.path.join(root, f"{i}.mmap"))
            names.append(f"{i}.mmap")
        reader = MelFileReader(root, clip_length=10)
        case = build_case(args, dev, 0, 1, "train", 625, B, 30)
        step = case["step"]
        for _ in range(2):
            reader.load_batch(names, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            xb = reader.load_batch(names, dev)
        torch.cuda.synchronize()
        t_load = (time.perf_counter() - t0) / 5
        # the step on a resident batch: the better of two passes (a busy host once made a single pass read 2-4 x the step)
        t_res = min(timed_steps(step, steps, 2, 1, dev), timed_steps(step, steps, 0, 1, dev)) / steps
        side = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=2)
        stop = threading.Event()

        def produce():
            torch.cuda.set_device(dev)
            while not stop.is_set():
                with torch.cuda.stream(side):
                    xb = reader.load_batch(names, dev)
                    ev = torch.cuda.Event()
                    ev.record(side)
                while not stop.is_set():
                    try:
                        q.put((xb, ev), timeout=0.05)
                        break
                    except queue.Full:
                        pass

        th = threading.Thread(target=produce, daemon=True)
        th.start()

        def fed_step():
            xb, ev = q.get()
            torch.cuda.current_stream().wait_event(ev)
            xb.record_stream(torch.cuda.current_stream())
            return step(xb)

        for _ in range(2):
            fed_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fed_step()
        torch.cuda.synchronize()
        t_fed = (time.perf_counter() - t0) / steps
        stop.set()
        th.join(timeout=5)
        hidden = max(0.0, min(1.0, 1.0 - (t_fed - t_res) / t_load))
        return {"what": "MelFileReader.load_batch (B float16 files on tmpfs -> pinned staging -> H2D -> melfile_assemble_kernel) "
                        "alone, and double-buffered on a second stream against the configs[2]-shaped training step (625 frames)",
                "per_gpu_batch": B, "files": "60 s raw float16 [3750, 96] on " + ("tmpfs (/dev/shm)" if root.startswith("/dev/shm") else "the temp dir"),
                "loader_alone_clips_per_s": round(B / t_load, 1), "loader_alone_ms_per_batch": round(t_load * 1e3, 2),
                "step_resident_ms": round(t_res * 1e3, 2), "step_resident_clips_per_s": round(B / t_res, 1),
                "step_fed_by_loader_ms": round(t_fed * 1e3, 2), "step_fed_by_loader_clips_per_s": round(B / t_fed, 1),
                "loader_over_step": round(t_res / t_load, 2),
                "loader_time_hidden_under_the_step": round(hidden, 3),
                "host_threads": 1}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        torch.cuda.empty_cache()


def side_case(args, dev, mode, T, B, patchout, steps, warmup, workload, precision=None, graph_too=False):
    """A further BASELINE configuration measured on the same line (N = 1 only): its own K timed steps between two
    synchronizes, then its own serialized kernel pass.  graph_too: the same K steps once more with the forward replayed from a
    captured HIP graph (BASELINE configs[4] names a hipGraph-captured forward), reported as `hip_graph_forward`."""
    if precision is not None:
        args = argparse.Namespace(**{**vars(args), "precision": precision})
    case = build_case(args, dev, 0, 1, mode, T, B, patchout)
    # Two brackets of K timed steps; `value` is the FIRST one, like the headline's single bracket (round 5 reported the faster of the
    # two: ADVICE r5), unless it is more than 25 % slower than the second -- the signature of the allocator stall a side case can catch
    # right after another configuration released tens of GB to the caching allocator (363 ms/step against a kernel sum of 83 once in ~7
    # default runs: profiles/r05d_default_line_boxes.txt) -- in which case the second is reported and `bracket_used` says so.  (The 30 s training
    # cases warm up for 6 steps: with four steps in flight the pool of a fresh 100 GB configuration is still growing after 2 -- four of five
    # round-6 boxes caught the stall in train30s's first bracket, profiles/r06_default_line_boxes.txt.)
    def alloc_counts():
        st = torch.cuda.memory_stats(dev)
        return [int(st.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")]
    a0 = alloc_counts()
    brackets = [timed_steps(case["step"], steps, warmup, 1, dev), timed_steps(case["step"], steps, 0, 1, dev)]
    a1 = alloc_counts()
    used = 1 if brackets[0] > 1.25 * brackets[1] else 0
    elapsed = brackets[used]
    step_flops, skipped, _, _ = flop_counts(case, args.precision)
    out = {"workload": workload, "value": round(B * steps / elapsed, 2), "unit": "clips/s", "steps": steps,
           "brackets_ms_per_step": [round(b / steps * 1e3, 3) for b in brackets],
           "bracket_used": "first" if used == 0 else "second (the first one was > 25 % slower: allocator stall)",
           # hipMalloc / hipFree calls and allocator retries of the caching allocator over the warm-up and both brackets (a hipFree drains the device)
           "device_alloc_free_retry": [a1[i] - a0[i] for i in range(3)],
           "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3), "per_gpu_batch": B, "mel": [96, T],
           "s_patchout_t": patchout, "tokens": case["N"], "dtype": args.precision,
           "model_tflops_per_s": round((step_flops - skipped) / (elapsed / steps) / 1e12, 1),
           "model_mfma_frac": round((step_flops - skipped) / (elapsed / steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
           "executed_flop_fraction": round(1.0 - skipped / step_flops, 4)}
    if mode == "infer":
        out["eval_streams"] = int(case["net"].eval_streams)   # 2: batches of >= MAEST.EVAL_SPLIT_ROWS token rows run as two halves on two streams
    if not args.no_kernel_timing:
        out.update(kernel_report(case, kernel_pass(case, steps), steps, args.precision, with_traffic=False))
    if args.precision != "fp32":
        try:
            out["deviation_vs_fp32"] = deviation_vs_fp32(case, args.precision)
        except Exception as e:  # pragma: no cover
            out["deviation_vs_fp32"] = {"error": repr(e)}
    if graph_too:
        try:
            case["net"].enable_hip_graph()
            tg = timed_steps(case["step"], steps, 3, 1, dev)      # (the warm-up steps capture)
            out["hip_graph_forward"] = {"value": round(B * steps / tg, 2), "unit": "clips/s", "ms_per_step": round(tg / steps * 1e3, 3),
                                        "steps": steps, "warmup": 3,
                                        "what": "the same K steps with the training forward replayed from a captured HIP graph "
                                                "(MAEST.enable_hip_graph; backward, loss and AdamW eager); the eager figure is `value`"}
            case["net"].enable_hip_graph(False)
        except Exception as e:  # pragma: no cover
            out["hip_graph_forward"] = {"error": repr(e)}
    del case
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 256 for the 10 s configs, 128 for ts)")
    ap.add_argument("--frames", type=int, default=None, help="mel frames per clip (10 s @ 16 kHz -> 626; 30 s -> 1876)")
    ap.add_argument("--patchout", type=int, default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "fp16"],
                    help="bf16: perf mode (the headline number); fp32: exact-fp32 MFMA parity mode; bf16x3: split-bf16 parity mode; "
                         "fp16: the perf mode's kernels on IEEE-half operands (the reference's 16-mixed arithmetic; training steps run "
                         "under torch.amp.GradScaler)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "ts"],
                    help="train: BASELINE configs[2] (the headline metric); infer: configs[1]; ts: configs[4] "
                         "(teacher-student, waveform -> HIP log-mel on the fly -> mixup -> 519-way separated heads, 30 s)")
    ap.add_argument("--hip-graph", action="store_true", help="replay the forward from a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=None, help="clips per oracle step of the CPU baseline (SURVEY 8d: B = 8)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-side-cases", action="store_true",
                    help="skip the `infer` / `infer_parity` (configs[1]), `train30s` (configs[3] per-GPU shape) and `ts` (configs[4]) "
                         "sub-objects of the default line")
    ap.add_argument("--complete-last-block", action="store_true",
                    help="evaluate the last block on every token (A/B reference for the head-token restriction)")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="disable the side-stream overlap of weight-gradient GEMMs (for kernel profiling)")
    ap.add_argument("--no-fold-delta", action="store_true",
                    help="attention backward: delta = rowsum(dO * O) by its own kernel instead of the proj dgrad GEMM's epilogue (A/B)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: create the one-rank RCCL communicator and push every gradient bucket through its "
                         "all-reduce anyway (the data-parallel exchange path on a single GPU)")
    ap.add_argument("--with-loader", action="store_true",
                    help="add the `loader` sub-object (the on-disk mel reader alone and double-buffered against the training step) "
                         "to a non-default line; the default line carries it anyway")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="debug / test switch: every rank of a --gpus N job runs on device 0 and gloo carries the device tensors of "
                         "the gradient exchange -- the whole N-rank launch, bootstrap, bucket and timing path on a one-GPU box "
                         "(tests/test_model_gpu.py); the clips/s of such a run measure nothing")
    ap.add_argument("--check-ranks", action="store_true",
                    help="after the timed steps: compare the weights of all ranks (they must be bit-identical) and report the "
                         "bucket all-reduces launched per step as `dp_check` on the line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

    from maest_amd.dist import init_from_env

    rank, local, world = init_from_env(backend="gloo" if args.ranks_share_gpu else None, force=args.force_collective)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.ranks_share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    ts = args.mode == "ts"
    train = args.mode in ("train", "ts")
    T = args.frames if args.frames is not None else (1876 if ts else 626)
    B = args.batch if args.batch is not None else (128 if T > 640 else 256)
    patchout = args.patchout if args.patchout is not None else ((90 if T > 640 else 30) if train else 0)
    case = build_case(args, dev, rank, world, args.mode, T, B, patchout)
    net, step, C, arch, N, Tk = case["net"], case["step"], case["C"], case["arch"], case["N"], case["Tk"]

    # ---- timed region: EXACTLY K steps, nothing but the steps between the two barriers/syncs
    elapsed = timed_steps(step, args.steps, args.warmup, world, dev)

    # ---- the same K steps with the last block evaluated on every token (N = 1 only; reported next to `value` so that the
    # effect of restricting it to the head's tokens is on the line itself)
    complete = None
    if world == 1 and getattr(net._engine, "head_tail", False) and not args.no_cpu_baseline and not args.hip_graph:
        net._engine.head_tail = False
        complete = timed_steps(step, args.steps, 2, 1, dev)
        net._engine.head_tail = True

    # ---- roofline pass (rank 0, N = 1)
    timer = None
    if not args.no_kernel_timing and world == 1:
        timer = kernel_pass(case, args.steps)

    # ---- data-parallel evidence on the line: per-bucket all-reduce times of the timed steps, and (always at N > 1) the rank check
    dp_buckets = None
    red = getattr(net, "_grad_sink", None)
    if red is not None and red.timing:
        dp_buckets = red.bucket_times()
    dp_check = None
    if args.check_ranks or world > 1:
        dp_check = check_ranks(net, world, dev)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        step_flops, skipped, attn_rows, tail_on = flop_counts(case, args.precision)
        secs = "10s" if T <= 640 else "30s"
        if ts:
            workload = ("discogs-maest-30s-pw-73e-ts teacher-student training step (BASELINE configs[4]): waveform -> "
                        "HIP log-mel on the fly -> mixup -> fwd (separated heads) -> (BCE + BCE)/2 -> bwd -> AdamW")
        elif train:
            workload = ("maest_10s_random_weights_pretrain training step (BASELINE configs[2]): mixup + fwd + BCE + bwd + AdamW"
                        if T <= 640 else f"30 s clips ({T} frames): maest_30s_from_passt_pretrain-shaped training step "
                        "(BASELINE configs[3], per-GPU shape)")
        else:
            workload = ("discogs-maest-10s-pw-129e inference (BASELINE configs[1])" if T <= 640
                        else f"30 s clips ({T} frames): discogs-maest-30s inference")
        collective = world > 1 or args.force_collective
        out = {
            "metric": f"clips/sec ({secs}@16kHz, 96-mel) MAEST-{secs} " + ("fwd+bwd" if train else "fwd"),
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload,
                       "arch": f"{arch} (DeiT-B distilled, 85.9M params), random init",
                       "per_gpu_batch": B, "global_batch": B * world,
                       "input": [B, (T - 1) * 256] if ts else [B, 96, T], "mel": [96, T],
                       "s_patchout_t": patchout, "tokens": N, "classes": C,
                       "hip_graph_forward": bool(args.hip_graph),
                       "parallelism": f"dp{world}" + (" (RCCL bucketed all-reduce, overlapped" +
                                                      ("; one-rank communicator, forced" if world == 1 else "") + ")"
                                                      if collective else "")},
            # FLOPs actually executed (the algorithmic count minus the last block's rows that feed nothing)
            "model_tflops_per_s": round((step_flops - skipped) * world / (elapsed / args.steps) / 1e12, 1),
            "model_mfma_frac": round((step_flops - skipped) / (elapsed / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "executed_flop_fraction": round(1.0 - skipped / step_flops, 4),
        }
        if dp_check is not None:
            out["dp_check"] = dp_check
        if dp_buckets:
            out["dp_buckets"] = {"what": "gradient buckets in backward order: size and launch -> complete time of their all-reduce on the "
                                         "communication stream, mean and max over the warm-up + timed steps (rank 0)",
                                 "buckets": dp_buckets, "wgrad_reserve_cus": int(net._engine.wgrad_reserve_cus),
                                 "sum_ms_per_step": round(sum(b["launch_to_complete_ms"] for b in dp_buckets), 3)}
        if args.ranks_share_gpu:
            out["config"]["parallelism"] += " -- ALL RANKS ON ONE GPU over gloo (--ranks-share-gpu: a path test, not a measurement)"
        if complete is not None:
            out["complete_last_block"] = {"value": round(B * args.steps / complete, 2), "unit": "clips/s",
                                          "ms_per_step": round(complete / args.steps * 1e3, 3),
                                          "what": "the same K steps with the last block evaluated on every token"}
        out["config"]["last_block"] = ("attention queries, proj, norm2 and MLP evaluated for the two tokens the head reads "
                                       "(cls, dist) only, forward and backward; logits, features and all gradients equal "
                                       "the complete evaluation's" if tail_on else "complete")
        if timer is not None:
            # HBM traffic of the dominant kernel from live PMC passes: on the headline configuration of a complete default
            # line only (N = 1, configs[2], bf16; MAEST_BENCH_PMC=0 or --no-cpu-baseline skip it: +50 s)
            live = None
            if (world == 1 and args.mode == "train" and T <= 640 and args.precision == "bf16" and not args.no_cpu_baseline
                    and B == 256 and os.environ.get("MAEST_BENCH_PMC", "1") != "0"):
                live = measure_pmc_traffic()
            out.update(kernel_report(case, timer, args.steps, args.precision, with_traffic=True, live_traffic=live))
        if world == 1 and not args.no_kernel_timing:
            # the HBM-bound front end of the path: the log-mel kernel on a full batch of waveforms
            try:
                out["mel"] = time_mel_kernel(dev, B, (T - 1) * 256)
            except Exception as e:  # pragma: no cover
                out["mel"] = {"error": repr(e)}
        if world == 1 and args.precision != "fp32":
            try:
                out["deviation_vs_fp32"] = deviation_vs_fp32(case, args.precision)
            except Exception as e:  # pragma: no cover
                out["deviation_vs_fp32"] = {"error": repr(e)}
        # ---- the other single-GPU BASELINE configurations, driver-visible on the same line (default run only)
        default_line = (world == 1 and args.mode == "train" and args.frames is None and args.batch is None
                        and args.patchout is None and not args.hip_graph and not args.no_side_cases
                        and not args.complete_last_block and not args.serial_kernels and not args.force_collective
                        and not args.no_fold_delta)
        if default_line:
            del case, net, step
            torch.cuda.empty_cache()
            try:
                out["infer"] = side_case(args, dev, "infer", 626, 256, 0, max(args.steps, 20), 3,
                                         "discogs-maest-10s-pw-129e inference (BASELINE configs[1]): batch 256 x (96 x 626) "
                                         "pre-extracted mel, N = 560 tokens -- north_star's '12-block attention at batch 256'")
            except Exception as e:  # pragma: no cover
                out["infer"] = {"error": repr(e)}
            try:
                out["infer_parity"] = side_case(args, dev, "infer", 626, 256, 0, 10, 2,
                                                "configs[1] in the mode that meets north_star's 1e-3 logits gate: precision "
                                                "\"bf16x3\" (fp32 tensors, three bf16 MFMAs per product; what a plain "
                                                "model.eval()(x) takes by default)", precision="bf16x3")
            except Exception as e:  # pragma: no cover
                out["infer_parity"] = {"error": repr(e)}
            try:
                out["infer_fp16"] = side_case(args, dev, "infer", 626, 256, 0, max(args.steps, 20), 3,
                                              "configs[1] in precision \"fp16\": the perf mode's kernels and schedules on IEEE-half operands "
                                              "(libmaest_hip_f16.so) -- the fast path INSIDE north_star's 1e-3 logits gate (`deviation_vs_fp32`)",
                                              precision="fp16")
            except Exception as e:  # pragma: no cover
                out["infer_fp16"] = {"error": repr(e)}
            try:
                out["train_fp16"] = side_case(args, dev, "train", 626, 256, 30, 10, 3,
                                              "the headline configuration (BASELINE configs[2]) in precision \"fp16\": graph recorded and differentiated on "
                                              "IEEE-half operands -- the reference's own GPU arithmetic (16-mixed, ex_maest.py:51) -- with torch.amp.GradScaler "
                                              "around AdamW as its trainer does (one host read of the inf / nan flag per step)", precision="fp16")
            except Exception as e:  # pragma: no cover
                out["train_fp16"] = {"error": repr(e)}
            try:
                out["ts"] = side_case(args, dev, "ts", 1876, 128, 90, 10, 6,
                                      "discogs-maest-30s-pw-73e-ts teacher-student training step (BASELINE configs[4], per-GPU "
                                      "shape): batch 128 x 30 s waveforms -> HIP log-mel on the fly -> mixup -> fwd (519-way "
                                      "separated heads) -> (BCE + BCE)/2 -> bwd -> AdamW", graph_too=True)
            except Exception as e:  # pragma: no cover
                out["ts"] = {"error": repr(e)}
            try:
                out["train30s"] = side_case(args, dev, "train", 1876, 128, 90, 10, 6,
                                            "maest_30s_from_passt_pretrain-shaped training step (BASELINE configs[3], the "
                                            "per-GPU shape of global batch 1024 over 8 GPUs): batch 128 x (96 x 1876), "
                                            "s_patchout_t 90, N = 875 tokens")
            except Exception as e:  # pragma: no cover
                out["train30s"] = {"error": repr(e)}
        if world == 1 and train and (default_line or args.with_loader):
            try:
                out["loader"] = loader_case(args, dev)
            except Exception as e:  # pragma: no cover
                out["loader"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                cb = args.cpu_batch if args.cpu_batch is not None else (8 if T <= 640 else 2)
                out["cpu_baseline"] = (cpu_baseline(cb, T, patchout, teacher_student=ts, waveform=ts) if train
                                       else cpu_baseline_infer(cb, T))
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(e)}
        # RCCL writes a version banner through C stdio, which (redirected to a file or pipe) would otherwise be flushed at
        # exit, BEHIND the JSON line: flush it out first so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # pragma: no cover
            pass
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
