"""pytest file (run AFTER tests/test_fullsize_gpu.py's inference tests in the same process, on a fresh box): the batch-256
fp32 training forward twice on the same inputs; every saved activation of the two passes must be bit-identical (the forward
has no atomics).  Prints the first tensor that differs and where."""
import numpy as np
import pytest
import torch
import sys
sys.path.insert(0, "tests")
import test_fullsize_gpu as T
from maest_amd.module import Module, _mix_to_device

pytestmark = pytest.mark.gpu


def test_first_forward_twice():
    net, _ = T._model("fp32", input_t=625, s_patchout_t=30)
    net.train()
    B, Tt = 256, 626
    x = T.randn((B, 1, 96, Tt), 21).to("cuda")
    rng = np.random.Generator(np.random.PCG64(22))
    y = torch.from_numpy((rng.random((B, 400)) < 0.00625).astype(np.float32)).to("cuda")
    Q = B // 4
    perm = torch.cat([torch.from_numpy(rng.permutation(Q)) + q * Q for q in range(4)])
    lam = torch.from_numpy(np.maximum(b := rng.beta(0.3, 0.3, B).astype(np.float32), 1 - b))
    Tp = (Tt - 16) // 10 + 1
    keep = torch.from_numpy(np.sort(rng.permutation(Tp)[: Tp - 30]))
    mod = Module(net=net, mixup_alpha=0.3)
    captured = []
    eng = net._engine
    orig = eng.forward

    def spy(*a, **k):
        outs, ctx = orig(*a, **k)
        snap = {}
        if ctx is not None:
            snap["cols"] = ctx["cols"].detach().clone()
            for i, s in enumerate(ctx["blocks"]):
                for kk, v in s.items():
                    if torch.is_tensor(v):
                        snap[f"blk{i}.{kk}"] = v.detach().clone()
        snap["logits"] = outs[0].detach().clone()
        captured.append(snap)
        return outs, ctx
    eng.forward = spy
    from maest_amd import ops as _ops
    pat = []
    orig_ta = _ops.token_assemble
    def ta_spy(patches, *a, **k):
        pat.append((patches.detach().clone(), patches.data_ptr(), [t.detach().clone() if torch.is_tensor(t) else t for t in a]))
        return orig_ta(patches, *a, **k)
    _ops.token_assemble = ta_spy
    losses = []
    for rep in range(3):
        for p in net.parameters():
            p.grad = None
        loss = mod.training_step((x, None, y), 0, _mixup=(perm, lam), _patchout=(0, keep))
        l_before = loss.item()
        loss.backward()
        torch.cuda.synchronize()
        losses.append((l_before, loss.item()))
    print("losses (before backward, after backward):", losses)
    _ops.token_assemble = orig_ta
    for rep in (1, 2):
        d = (pat[0][0] - pat[rep][0]).abs()
        idx = torch.nonzero(d.amax(1) > 0).flatten()
        print(f"pass 0 vs {rep}: patches (GEMM out) max diff {d.max().item():.3e} rows {idx[:8].tolist()} n={idx.numel()}  ptrs {pat[0][1]:#x} {pat[rep][1]:#x}")
        for j, (a0, a1) in enumerate(zip(pat[0][2], pat[rep][2])):
            if torch.is_tensor(a0) and not torch.equal(a0, a1):
                print(f"pass 0 vs {rep}: token_assemble arg {j} differs, max {(a0.float() - a1.float()).abs().max().item():.3e} shape {tuple(a0.shape)}")
    bad = False
    for rep in (1, 2):
        for k in captured[0]:
            a, b2 = captured[0][k], captured[rep][k]
            if not torch.equal(a, b2):
                d = (a.float() - b2.float()).abs()
                idx = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten()
                print(f"pass 0 vs {rep}: {k} differs: max {d.max().item():.3e}, rows {idx[:10].tolist()} ... ({idx.numel()} rows of {d.shape[0]})")
                bad = True
                n_shown = locals().get("n_shown", 0) + 1
                if n_shown >= 6:
                    break
    assert not bad and all(abs(a - b) < 1e-9 for a, b in losses)
