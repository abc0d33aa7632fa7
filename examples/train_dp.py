#!/usr/bin/env python
"""Data-parallel MAEST training loop on MI355X: on-disk float16 mel chunks -> device input (MelFileReader), fused
SpecMasking + mixup + patchout + ViT step (Module.training_step), bucketed RCCL gradient all-reduce (GradReducer),
AdamW, SWA.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_dp.py --data DIR
    python examples/train_dp.py            # single GPU, synthetic mel files written to a temp dir

What replaces what: discogs/dataset.py + datamodule.py loader workers -> MelFileReader; helpers/spec_masking.py (applied
per clip by those workers) -> SpecMasking stripes drawn per step and applied inside the patch-embedding operand load;
Lightning DDP -> GradReducer; helpers/swa_callback.py -> WeightAverager; models/module.py:Module -> maest_amd.module.Module.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maest_amd.dist import GradReducer, broadcast_parameters, init_from_env  # noqa: E402
from maest_amd.melfile import MelFileReader  # noqa: E402
from maest_amd.module import Module  # noqa: E402
from maest_amd.spec_masking import SpecMasking  # noqa: E402
from maest_amd.swa import WeightAverager  # noqa: E402


def synthetic_dataset(root, n=64, classes=400):
    rng = np.random.Generator(np.random.PCG64(0))
    names = []
    for i in range(n):
        frames = int(rng.integers(400, 2000))
        (rng.random((frames, 96), dtype=np.float32) * 4.0).astype("float16").tofile(os.path.join(root, f"{i}.mel"))
        names.append(f"{i}.mel")
    y = (rng.random((n, classes)) < 0.01).astype(np.float32)
    return names, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per GPU")
    ap.add_argument("--no-spec-masking", action="store_true")
    ap.add_argument("--hip-graph", action="store_true", help="replay the training forward from a captured HIP graph")
    ap.add_argument("--precision", default="auto", choices=["auto", "bf16", "fp16", "fp32"],
                    help="fp16: the reference's 16-mixed arithmetic (IEEE-half operands) -- the loop then runs under torch.amp.GradScaler like its trainer")
    args = ap.parse_args()
    rank, local, world = init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    tmp = None
    root = args.data
    if root is None:
        tmp = tempfile.TemporaryDirectory()
        root = tmp.name
    names, y_all = synthetic_dataset(root) if args.data is None else (sorted(os.listdir(root)), None)
    if y_all is None:
        y_all = np.zeros((len(names), 400), np.float32)       # plug the ground-truth pickle of the reference here

    mod = Module(arch="passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30,
                 spec_masking=None if args.no_spec_masking else SpecMasking()).to(dev)
    mod.net.precision = args.precision
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14) if args.precision == "fp16" else None    # gradients in half need a scaled loss
    mod.net.train()
    if args.hip_graph:
        mod.net.enable_hip_graph()
    broadcast_parameters(mod.net)
    cfg = mod.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]        # exp warm-up / linear ramp-down, stepped per "epoch"
    reducer = None
    if world > 1:
        reducer = GradReducer(mod.net.named_parameters(), skip=("head_dist.weight", "head_dist.bias"))
        mod.net._grad_sink = reducer
    reader = MelFileReader(root, clip_length=10)
    swa = WeightAverager(mod.net)
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    for step in range(args.steps):
        idx = rng.integers(0, len(names), args.batch)
        x = reader.load_batch([names[i] for i in idx], dev)                 # pad / roll / transpose / normalise on the GPU
        y = torch.from_numpy(y_all[idx]).to(dev)
        if reducer is not None:
            reducer.reset()
        loss = mod.training_step((x, None, y), step)
        (loss if scaler is None else scaler.scale(loss)).backward()
        if reducer is not None:
            reducer.finish()
        if scaler is None:
            opt.step()
        else:
            scaler.step(opt)               # unscales, skips the step on inf / nan
            scaler.update()
        opt.zero_grad(set_to_none=reducer is None)
        if (step + 1) % 10 == 0:          # this toy run calls 10 steps an epoch
            sched.step()
            swa.update()
            if rank == 0:
                print(f"step {step + 1}: loss {loss.item():.4f}  (SWA over {swa.n_averaged} snapshots)")
    if rank == 0:
        torch.save(swa.state_dict(), os.path.join(root, "last.ckpt"))         # loads with get_maest(checkpoint=...)
        print("saved", os.path.join(root, "last.ckpt"))
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
