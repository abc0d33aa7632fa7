#!/usr/bin/env python
"""Generate tests/golden/g12_patch_stride.npz from the IMPORTED reference (run in the authoring container only).

TEST INFRASTRUCTURE.  The reference takes the patch-embedding strides as constructor arguments (get_maest(stride_f=, stride_t=):
models/maest.py:1505-1507, 1537; PatchEmbed: models/maest.py:214-241).  Every published architecture uses (10, 10); this fixture pins
another pair -- (16, 13): six frequency rows, the frequency table of 96 // 16 = 6 entries fits them -- for an evaluation forward and for a
training forward with structured patchout and a time-table offset, plus what the reference does with a stride whose frequency table does
NOT fit the patch grid (stride_f = 8: grid 96 // 8 = 12 against 11 rows of patches): the error class and message are recorded.

    python oracle/gen_golden_stride.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from oracle import maest_oracle as O  # noqa: E402
from gen_golden import OUT, check, import_reference, randn  # noqa: E402

STRIDE = (16, 13)


def main():
    torch.set_num_threads(8)
    rm = import_reference()
    m = rm.get_maest("discogs-maest-10s-pw-129e", pretrained=False, stride_f=STRIDE[0], stride_t=STRIDE[1])
    sd = O.make_state_dict(625, n_classes=m.num_classes, stride=STRIDE)
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = randn((2, 96, 626), 71)
    with torch.no_grad():
        logits, feats = m(x.clone())
        ol, of = O.forward(x.clone(), sd, (96, 625), stride=STRIDE)
    check("G12 logits", ol, logits)
    check("G12 feats", of, feats)

    # training forward: structured time patchout of 7 columns and the random time-table offset, both captured from the reference's draws
    mt = rm.get_maest("discogs-maest-10s-pw-129e", pretrained=False, stride_f=STRIDE[0], stride_t=STRIDE[1], s_patchout_t=7)
    mt.load_state_dict(sd, strict=True)
    mt.train()
    xs = randn((2, 96, 500), 72)                       # shorter than the table: an offset is drawn
    Tp = (500 - 16) // STRIDE[1] + 1
    table = sd["time_new_pos_embed"].shape[-1]
    torch.manual_seed(5)
    toffset = int(torch.randint(1 + table - Tp, (1,)).item())
    t_keep = torch.randperm(Tp)[: Tp - 7].sort().values.tolist()
    torch.manual_seed(5)
    with torch.no_grad():
        lt, ft = mt(xs.clone())
        olt, oft = O.forward(xs.clone(), sd, (96, 625), toffset=toffset, t_keep=t_keep, stride=STRIDE)
    check("G12 train logits", olt, lt)

    # a stride whose frequency table does not fit the patch grid
    mb = rm.get_maest("discogs-maest-10s-pw-129e", pretrained=False, stride_f=8, stride_t=10)
    mb.eval()
    err_type, err_msg = "", ""
    try:
        with torch.no_grad():
            mb(x.clone())
    except Exception as e:  # noqa: BLE001
        err_type, err_msg = type(e).__name__, str(e)
    print(f"  stride_f = 8: {err_type}: {err_msg}")
    assert err_type, "the reference accepted stride_f = 8"
    np.savez(os.path.join(OUT, "g12_patch_stride.npz"), stride=np.array(STRIDE), logits=logits.numpy(), features=feats.numpy(),
             train_logits=lt.numpy(), train_features=ft.numpy(), toffset=np.array(toffset), t_keep=np.array(t_keep),
             bad_stride_error_type=np.array(err_type), bad_stride_error_message=np.array(err_msg))
    print("wrote g12_patch_stride.npz")


if __name__ == "__main__":
    main()
