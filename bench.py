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
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def flops_per_clip_fwd(N, Tk, C=400):
    per_block = 2 * N * 768 * (2304 + 768 + 3072 + 3072) + 4 * N * N * 768
    return 12 * per_block + 2 * (9 * Tk) * 256 * 768 + 2 * 768 * C


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot be collected by bench.py on itself); None when the file is absent."""
    path = os.path.join(REPO, "profiles", "r01g_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline_infer(batch, T, steps=2):
    """Oracle eval forward on the host cores (B = 8, N = 560)."""
    from oracle import maest_oracle as O
    sd = O.make_state_dict(625, seed=1234)
    rng = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(rng.standard_normal((batch, 96, T), dtype=np.float32))
    times = []
    with torch.no_grad():
        for it in range(steps + 1):
            t0 = time.perf_counter()
            O.forward(x, sd, (96, 625), melspectrogram_input=True)
            if it > 0:
                times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"value": round(batch / t, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle eval forward (fp32) batch={batch} T={T}; median of {steps} passes after 1 warm-up, "
                      f"{t:.2f} s/pass"}


def cpu_baseline(batch, T, patchout, steps=2):
    """Oracle training step (fwd + bwd + AdamW) on the host cores: the reference's algorithm."""
    from oracle import maest_oracle as O
    torch.manual_seed(0)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.make_state_dict(625, seed=1234).items()}
    params = [v for k, v in sd.items() if not k.startswith("head_dist")]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=1e-4)
    rng = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(rng.standard_normal((batch, 1, 96, T), dtype=np.float32))
    y = torch.from_numpy((rng.random((batch, 400)) < 0.00625).astype(np.float32))
    Tp = (T - 16) // 10 + 1
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        perm = torch.randperm(batch)
        lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, batch).astype(np.float32), 1 - b))
        keep = torch.randperm(Tp)[: Tp - patchout].sort().values.tolist()
        loss, _ = O.training_loss(x, y, sd, perm, lam, toffset=0, t_keep=keep)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if it > 0:
            times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    return {"value": round(batch / t, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle training step (fwd+bwd+AdamW, fp32) batch={batch} T={T} patchout={patchout}; "
                      f"median of {steps} steps after 1 warm-up, {t:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config 3: 256)")
    ap.add_argument("--frames", type=int, default=626, help="mel frames per clip (10 s @ 16 kHz -> 626)")
    ap.add_argument("--patchout", type=int, default=30)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="clips per oracle step of the CPU baseline (SURVEY 8d: B = 8)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--serial-kernels", action="store_true",
                    help="disable the side-stream overlap of weight-gradient GEMMs (for kernel profiling)")
    args = ap.parse_args()

    from maest_amd import get_maest, ops
    from maest_amd.dist import GradReducer, broadcast_parameters, init_from_env
    from maest_amd.module import Module

    rank, local, world = init_from_env()
    if world != args.gpus:
        if args.gpus > 1 and world == 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} ...`")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    B, T = args.batch, args.frames
    train = args.mode == "train"
    net = get_maest("passt_s_swa_p16_128_ap476" if train else "discogs-maest-10s-pw-129e", pretrained=False,
                    input_t=(T // 5) * 5 if T > 640 else 625,   # time table: 62 columns for 10 s, 187 for the 30 s configs
                    s_patchout_t=args.patchout if train else 0, precision=args.precision).to(dev)
    broadcast_parameters(net)
    if args.serial_kernels:
        net._engine.overlap_wgrad = False
    mod = Module(net=net, mixup_alpha=0.3)
    Tp = (T - 16) // 10 + 1
    Tk = Tp - (args.patchout if train else 0)
    N = 2 + 9 * Tk

    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    x = torch.randn((B, 1, 96, T), generator=gen, device=dev)                   # synthetic z-normed log-mel
    y = (torch.rand((B, 400), generator=gen, device=dev) < 0.00625).float()     # ~2.5 labels per clip

    if train:
        net.train()
        opt = mod.configure_optimizers()
        reducer = None
        if world > 1:
            reducer = GradReducer(net.named_parameters(), skip=("head_dist.weight", "head_dist.bias"))
            net._grad_sink = reducer

        def step():
            if reducer is not None:
                reducer.reset()
            loss = mod.training_step((x, None, y), 0)
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            opt.zero_grad(set_to_none=reducer is None)
            return loss
    else:
        net.eval()

        def step():
            with torch.no_grad():
                return net(x)[0]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # ---- timed region: EXACTLY K steps, nothing but the steps between the two barriers/syncs
    t0 = time.perf_counter()
    for _ in range(args.steps):
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

    # ---- roofline pass (rank 0, N = 1): the same K steps again with HIP events around every launch of the
    # GEMM / attention kernels on the launch stream.  Kept out of the timed region above because ~400 event
    # records per step cost ~10 % of a step on the host; kernel durations themselves are unaffected.
    timer = None
    if not args.no_kernel_timing and world == 1:
        # kernels are timed one at a time: with the wgrad GEMMs overlapping the dgrad chain on a second
        # stream, an event pair around one launch would also count the time it shares the CUs with another
        net._engine.overlap_wgrad = False
        with ops.KernelTimer(kinds={"maest_gemm_nt", "maest_gemm_tn", "maest_attn_fwd", "maest_attn_bwd"}) as timer:
            for _ in range(args.steps):
                step()
        torch.cuda.synchronize()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        fwd = flops_per_clip_fwd(N, Tk)
        step_flops = (3 if train else 1) * fwd * B
        out = {
            "metric": ("clips/sec (10s@16kHz, 96-mel) MAEST-10s " if T <= 640 else "clips/sec (30s@16kHz, 96-mel) MAEST-30s ")
                      + ("fwd+bwd" if train else "fwd"),
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (("maest_10s_random_weights_pretrain training step (BASELINE configs[2]): "
                                     "mixup + fwd + BCE + bwd + AdamW" if train else
                                     "discogs-maest-10s-pw-129e inference (BASELINE configs[1])") if T <= 640 else
                                    (f"30 s clips ({T} frames): " + ("maest_30s_from_passt_pretrain-shaped training step "
                                     "(BASELINE configs[3], per-GPU shape)" if train else "discogs-maest-30s inference"))),
                       "arch": "passt_s_swa_p16_128_ap476 (DeiT-B distilled, 85.9M params), random init",
                       "per_gpu_batch": B, "global_batch": B * world, "mel": [96, T],
                       "s_patchout_t": args.patchout if train else 0, "tokens": N, "classes": 400,
                       "parallelism": f"dp{world}" + (" (RCCL bucketed all-reduce, overlapped)" if world > 1 else "")},
            "model_tflops_per_s": round(step_flops * world / (elapsed / args.steps) / 1e12, 1),
            "model_mfma_frac": round(step_flops / (elapsed / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        if timer is not None:
            summ = timer.summary()
            g = summ.get("maest_gemm_nt")
            if g and g["ms"] > 0:
                ach = g["work"] / (g["ms"] * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "kernel": ("maest_gemm_nt (gemm_nt256w_kernel<bf16>; gemm_nt_kernel<bf16> for the 2 small head GEMMs)"
                                              if args.precision == "bf16" else "maest_gemm_nt (fp32 MFMA)"),
                                   "achieved": round(ach, 1),
                                   "peak": PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3,
                                   "unit": "TFLOP/s",
                                   "frac": round(ach / (PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3), 4),
                                   "traffic": pmc_traffic(),
                                   "launches_per_step": g["launches"] // args.steps,
                                   "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                                   "ms_per_step": round(g["ms"] / args.steps, 3),
                                   "note": "second pass over the same K steps, kernels serialized (side-stream "
                                           "overlap off) so that each event pair times one kernel alone"}
            out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in summ.items()}
            att = [summ.get("maest_attn_fwd"), summ.get("maest_attn_bwd")]
            aw = sum(a["work"] for a in att if a)
            am = sum(a["ms"] for a in att if a)
            if am > 0:
                out["attention_core_tflops"] = round(aw / (am * 1e-3) / 1e12, 1)
            # north_star: "fraction of the attention-GEMM roofline" = the 12-block attention set (QKV projection +
            # QK^T + PV + output projection), its launches picked out of the timed records by their
            # algorithmic work (the MLP GEMMs have 3072-wide shapes)
            Mtok = B * N
            set_work = {2.0 * Mtok * 2304 * 768, 2.0 * Mtok * 768 * 768}
            set_ms = sum(e0.elapsed_time(e1) for name, e0, e1, w in timer.records
                         if name in ("maest_attn_fwd", "maest_attn_bwd")
                         or (name in ("maest_gemm_nt", "maest_gemm_tn") and w in set_work)) / args.steps
            set_flops = (3 if train else 1) * B * 12 * (2.0 * N * 768 * 2304 + 4.0 * N * N * 768 + 2.0 * N * 768 * 768)
            if set_ms > 0:
                out["attention_set"] = {"what": "12 x (QKV proj + QK^T + PV + out proj)" + (", fwd+bwd" if train else ", fwd"),
                                        "ms_per_step": round(set_ms, 3), "tflops": round(set_flops / set_ms / 1e9, 1),
                                        "mfma_frac": round(set_flops / set_ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = (cpu_baseline(args.cpu_batch, T, args.patchout) if train
                                       else cpu_baseline_infer(args.cpu_batch, T))
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
