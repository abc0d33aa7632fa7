#!/usr/bin/env python
"""Tag one audio clip (or a synthetic one) with MAEST on an MI355X -- the reference's README usage, unchanged but
for `.cuda()`:

    python examples/infer.py [--seconds 35] [--arch discogs-maest-30s-pw-129e] [--checkpoint last.ckpt]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maest import get_maest  # noqa: E402  (alias package of maest_amd, as in the reference)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="discogs-maest-30s-pw-129e")
    ap.add_argument("--seconds", type=float, default=35.0)
    ap.add_argument("--checkpoint", default=None, help="Lightning .ckpt with net_swa.* / net.* weights")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "bf16", "bf16x3", "fp16"])
    args = ap.parse_args()

    model = get_maest(args.arch, pretrained=False, checkpoint=args.checkpoint, precision=args.precision).cuda().eval()
    rng = np.random.Generator(np.random.PCG64(0))
    audio = torch.from_numpy((rng.standard_normal(int(args.seconds * 16000)) * 0.1).astype(np.float32)).cuda()

    with torch.no_grad():
        logits, embeddings = model(audio)                       # 1-D audio: mel on the GPU, chunked into a batch
        _, emb7 = model(audio, transformer_block=6)             # cls + dist + mean token of block 6
        activations, labels = model.predict_labels(audio)
    top = np.argsort(activations)[::-1][:5]
    print("logits", tuple(logits.shape), "embeddings", tuple(embeddings.shape), "block-6 embedding", tuple(emb7.shape))
    for i in top:
        print(f"  {activations[i]:.3f}  {labels[i]}")


if __name__ == "__main__":
    main()
