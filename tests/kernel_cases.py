"""Kernel-level parity cases shared by the emulator tests (CPU, tiny shapes) and the GPU tests.

Every case runs ONE C-ABI entry point through maest_amd.ops and compares it with the oracle
(oracle/maest_oracle.py) or with the one-line torch definition of the op, in fp32 on the CPU.
Tolerances: fp32 ("parity") mode 2e-5 relative unless stated; bf16 mode is checked against the
same math evaluated on bf16-ROUNDED operands (so the only difference is accumulation order).
"""
import math

import numpy as np
import torch
from functools import partial
import torch.nn.functional as F

from maest_amd import _lib, ops
from oracle import maest_oracle as O


def rnd(shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * np.float32(scale))


def close(a, b, rtol, atol, what):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    bad = err > lim
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; max err {err.max().item():.3e} "
        f"(ref max {b.abs().max().item():.3e}); first bad index {tuple(int(i) for i in bad.nonzero()[0])}")


def tol(dtype):
    return (2e-5, 2e-5) if dtype == torch.float32 else (2e-2, 2e-2)


# ------------------------------------------------------------------------------------------ GEMM
def case_gemm(dev, dtype, M, N, K, seed=0, identity=True):
    a = rnd((M, K), seed).to(dtype)
    b = rnd((N, K), seed + 1).to(dtype)
    bias = rnd((N,), seed + 2)
    ref = a.float() @ b.float().t() + bias
    # fp32 accumulation-order noise of a K-term dot product of N(0,1) operands: ~ 4e-7 * K absolute
    rt, at = 1e-5, 4e-7 * K / math.sqrt(K / 64)
    c = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_dtype=torch.float32)
    close(c, ref, rt, at * math.sqrt(K / 64), "gemm none")
    # asymmetric A = I check of the output orientation
    if identity and M >= K and dtype == torch.float32:
        eye = torch.zeros(M, K)
        eye[:K, :K] = torch.eye(K)
        c = ops.gemm_nt(eye.to(dev), b.to(dev), None, out_dtype=torch.float32)
        close(c[:K], b.float().t(), 0, 1e-6, "gemm identity (transpose-detecting)")
    # GELU epilogue with aux_out
    aux = torch.empty((M, N), dtype=dtype, device=dev)
    g = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_dtype=dtype, epi=ops.EPI_GELU, aux_out=aux)
    rt2, at2 = (1e-5, at) if dtype == torch.float32 else (1e-2, 1e-2)
    xr = ref.clone().requires_grad_(True)
    F.gelu(xr).sum().backward()
    close(aux, xr.grad, rt2, max(at2 * math.sqrt(K / 64), 2e-6), "gemm gelu aux (= gelu' saved for backward)")
    close(g, F.gelu(ref), rt2, at2 * math.sqrt(K / 64), "gemm gelu")
    # residual epilogue
    res = rnd((M, N), seed + 3)
    c = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_dtype=torch.float32, epi=ops.EPI_RESIDUAL,
                    aux_in=res.to(dev))
    close(c, ref + res, rt, at * math.sqrt(K / 64), "gemm residual")
    # mul epilogue (dgrad through GELU: acc * saved gelu')
    pre = rnd((M, N), seed + 4).to(dtype)
    c = ops.gemm_nt(a.to(dev), b.to(dev), None, out_dtype=dtype, epi=ops.EPI_MUL, aux_in=pre.to(dev))
    close(c, (ref - bias) * pre.float(), rt2, at2 * math.sqrt(K / 64), "gemm mul")
    # split-K atomic accumulate
    acc = torch.zeros((M, N), dtype=torch.float32, device=dev)
    ops.gemm_nt(a.to(dev), b.to(dev), None, out=acc, epi=ops.EPI_ATOMIC, split_k=3)
    close(acc, ref - bias, rt, at * math.sqrt(K / 64), "gemm split-k")


def _same_products(new, old, exact, what, K=64):
    """The one-wave-per-SIMD GEMM against the 8-wave kernel on the same operands.  `exact` (the host emulator, whose MFMA twins add the k terms
    one by one in both shapes): bit for bit.  On the device the two kernels use different matrix instructions since round 6 (16x16x32 against
    32x32x16: 32 against 16 products per rounding step), so their fp32 sums differ in the last bits: fp32 outputs within 4e-7 sqrt(K) of the output
    scale, bf16 outputs within one bf16 ulp and at most 2 % of the elements different at all."""
    if exact:
        assert torch.equal(new, old), what
        return
    a, b = new.float(), old.float()
    scale = float(b.abs().max()) + 1e-30
    diff = (a - b).abs()
    if new.dtype == torch.float32:
        assert float(diff.max()) <= 4e-7 * math.sqrt(K) * scale + 1e-30, f"{what}: max |diff| {float(diff.max()):.3e} of scale {scale:.3e}"
    else:
        ulp = 2.0 ** -7 * torch.maximum(a.abs(), b.abs()) + 1e-30          # one bf16 ulp is 2^-8 .. 2^-7 of the value
        assert bool((diff <= ulp).all()), f"{what}: more than one bf16 ulp apart (max ratio {float((diff / ulp).max()):.2f})"
        assert float((diff > 0).float().mean()) <= 0.02, f"{what}: {100 * float((diff > 0).float().mean()):.2f} % of the elements differ"


def case_gemm_one_wave_per_simd(dev, M, N, K, seed=11, only=None, pair=True, both_bias=False, exact=None):
    """gemm_nt256o_kernel (gemm_nt_ow.hip: bf16 operands, the default of the 256 x 256 path) against the 8-wave kernel it replaces
    (gemm_variant = 3): the same products summed in the same order and the same epilogue arithmetic -- bit for bit, in every
    epilogue form, ragged last tile row included -- and against the oracle's fp32 matmul."""
    dt = torch.bfloat16
    a = rnd((M, K), seed).to(dt).to(dev)
    w = (rnd((N, K), seed + 1) * 0.1).to(dt).to(dev)
    bias = rnd((N,), seed + 2).to(dev)
    res = rnd((M, N), seed + 3).to(dev)
    mul = rnd((M, N), seed + 4).to(dt).to(dev)
    ref = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    forms = [("none -> bf16", dict(out_dtype=dt)), ("none -> fp32", dict(out_dtype=torch.float32)),
             ("gelu -> bf16", dict(out_dtype=dt, epi=ops.EPI_GELU)), ("gelu -> fp32", dict(out_dtype=torch.float32, epi=ops.EPI_GELU)),
             ("residual -> fp32", dict(out_dtype=torch.float32, epi=ops.EPI_RESIDUAL, aux_in=res)),
             ("mul -> bf16", dict(out_dtype=dt, epi=ops.EPI_MUL, aux_in=mul))]
    for name, kw in forms:
        if only is not None and name not in only:
            continue
        for b in ((bias, None) if (only is None or both_bias) else (bias,)):
            new = ops.gemm_nt(a, w, b, **kw)
            with ops.options(gemm_variant=3):
                old = ops.gemm_nt(a, w, b, **kw)
            _same_products(new, old, str(dev) == "cpu" if exact is None else exact,
                           f"one-wave-per-SIMD GEMM differs from the 8-wave kernel: {name}, bias {b is not None}", K)
    c = ops.gemm_nt(a, w, bias, out_dtype=torch.float32)
    close(c, ref, 1e-5, 4e-7 * K, "one-wave-per-SIMD GEMM vs fp32 matmul")
    if not pair:
        return
    aux_n = torch.empty((M, N), dtype=dt, device=dev)
    aux_o = torch.empty((M, N), dtype=dt, device=dev)
    g_n = ops.gemm_nt(a, w, bias, out_dtype=dt, epi=ops.EPI_GELU, aux_out=aux_n)
    with ops.options(gemm_variant=3):
        g_o = ops.gemm_nt(a, w, bias, out_dtype=dt, epi=ops.EPI_GELU, aux_out=aux_o)
    _same_products(g_n, g_o, str(dev) == "cpu", "GELU of the pair form differs between the two 256 x 256 kernels", K)
    _same_products(aux_n, aux_o, str(dev) == "cpu", "GELU' of the pair form differs between the two 256 x 256 kernels", K)
    close(g_n, F.gelu(ref), 1e-2, 1e-2 * math.sqrt(K / 64), "one-wave-per-SIMD GEMM: gelu")


def case_gemm_rowdot(dev, dtype, M, N, K, ntok, seed=7):
    """maest_gemm_nt_rowdot: C = A B^T + bias in `dtype`, and rowdot[item, 64-column group, row in item] = the dot
    product of the STORED row segment of C with `other` -- the attention backward's delta out of the dgrad GEMM's
    epilogue.  Reference: the products of the returned C itself (exactly the values the kernel multiplied)."""
    a = rnd((M, K), seed).to(dtype)
    b = rnd((N, K), seed + 1).to(dtype)
    bias = rnd((N,), seed + 2)
    other = rnd((M, N), seed + 3).to(dtype)
    c, rd = ops.gemm_nt_rowdot(a.to(dev), b.to(dev), other.to(dev), ntok, out_dtype=dtype, bias=bias.to(dev))
    ref = a.float() @ b.float().t() + bias
    rt, at = (1e-5, 4e-7 * K) if dtype == torch.float32 else (1e-2, 1e-2 * math.sqrt(K / 64))
    close(c, ref, rt, at, "gemm rowdot: C")
    assert rd.shape == (M // ntok, N // 64, ntok)
    want = (c.float().cpu() * other.float()).reshape(M // ntok, ntok, N // 64, 64).sum(-1).permute(0, 2, 1)
    close(rd, want, 1e-5, 1e-5 * math.sqrt(64) * float(c.float().abs().max()), "gemm rowdot: per-(row, group) dot products")
    if dtype == torch.bfloat16:
        # bf16: the call above took gemm_nt256o_kernel where the shape has 256-row tiles; the 8-wave kernel's row-dot epilogue does the
        # same arithmetic in the same order
        with ops.options(gemm_variant=3):
            c3, rd3 = ops.gemm_nt_rowdot(a.to(dev), b.to(dev), other.to(dev), ntok, out_dtype=dtype, bias=bias.to(dev))
        _same_products(c, c3, str(dev) == "cpu", "row-dot epilogue: C differs between the one-wave-per-SIMD and 8-wave kernels", K)
        close(rd, rd3, 1e-2, 1e-2 * math.sqrt(64) * float(c.float().abs().max()) * 2.0 ** -7, "row-dot epilogue: dot products of the two kernels")


def case_gemm_tn(dev, dtype, K, M, N, seed=3, lda_pad=0, splits=(1, 3, 0)):
    """wgrad form: out[M,N] += a[K,M]^T b[K,N], colsum[M] += a.sum(0); ragged K (token tail)."""
    a_full = rnd((K, M + lda_pad), seed).to(dtype)
    a = a_full[:, :M]
    b = rnd((K, N), seed + 1).to(dtype)
    ref = a.float().t() @ b.float()
    ref_cs = a.float().sum(0)
    at = 4e-7 * K + (0 if dtype == torch.float32 else 1e-3)
    for sk in splits:
        out = torch.zeros((M, N), dtype=torch.float32, device=dev)
        cs = torch.zeros(M, dtype=torch.float32, device=dev)
        a_dev = a_full.to(dev)[:, :M]
        ops.gemm_tn(a_dev, b.to(dev), out, colsum=cs, split_k=sk, M=M, N=N)
        close(out, ref, 1e-5, at, f"gemm_tn split_k={sk}")
        close(cs, ref_cs, 1e-5, at, f"gemm_tn colsum split_k={sk}")
    # the deterministic split-K combine of the 256-tile kernel (tn_reduce=1; where the shape takes it): partial tiles through a
    # workspace, summed in split order by a second kernel -- bit-reproducible, and ACCUMULATING into `out` like the atomics
    a_dev, b_dev = a_full.to(dev)[:, :M], b.to(dev)
    with ops.options(tn_reduce=1):
        outs = []
        for _ in range(1 if _lib.host_emulation() else 2):
            out = torch.zeros((M, N), dtype=torch.float32, device=dev)
            ops.gemm_tn(a_dev, b_dev, out, split_k=0, M=M, N=N)
            outs.append(out)
        close(outs[0], ref, 1e-5, at, "gemm_tn (workspace combine)")
        if ops.gemm_tn_workspace_bytes(dtype, M, N, K) > 0 and len(outs) == 2:
            assert torch.equal(outs[0], outs[1]), "gemm_tn through the workspace must be bit-reproducible"
            ops.gemm_tn(a_dev, b_dev, outs[1], split_k=0, M=M, N=N)            # second call accumulates
            close(outs[1], 2 * ref, 1e-5, 2 * at, "gemm_tn accumulates into a non-zero C")
    # transpose-detecting: A = [I | 0] picks rows of B
    if dtype == torch.float32 and K >= M:
        eye = torch.zeros(K, M)
        eye[:M, :M] = torch.eye(M)
        out = torch.zeros((M, N), dtype=torch.float32, device=dev)
        ops.gemm_tn(eye.to(dev), b.to(dev), out)
        close(out, b.float()[:M], 0, 1e-6, "gemm_tn identity")


# ------------------------------------------------------------------------------------- transposes
def case_transpose(dev, dtype, rows, cols):
    src = rnd((rows, cols), 5).to(dtype)
    ld = ops.round_up(rows, 64)
    out = ops.transpose(src.to(dev), ld)
    assert out.shape == (cols, ld)
    close(out[:, :rows], src.t(), 0, 0, "transpose")
    assert float(out[:, rows:].float().abs().sum()) == 0.0, "transpose pad must be zero"
    w = rnd((rows, cols), 6)
    d, dt_ = ops.cast_weights(w.to(dev), dtype, want=True, want_t=True)
    close(d, w.to(dtype), 0, 0, "cast")
    close(dt_, w.to(dtype).t(), 0, 0, "cast transposed")
    # many parameters, one launch (ragged shapes, NULL outputs)
    # (sides that are multiples of 4 take the kernel's quad path -- 16-byte loads, 8-byte stores --, the others the element-wise one)
    ws = [rnd((rows, cols), 60), rnd((cols, 33), 61), rnd((65, 64), 62), rnd((132, 72), 63), rnd((64, 256), 64)]
    for want, want_t in ((True, True), (False, True), (True, False)):
        outs = ops.cast_weights_multi([t.to(dev) for t in ws], dtype, want=want, want_t=want_t)
        for t, (o, ot) in zip(ws, outs):
            assert (o is None) == (not want) and (ot is None) == (not want_t)
            if o is not None:
                close(o, t.to(dtype), 0, 0, "cast multi")
            if ot is not None:
                close(ot, t.to(dtype).t(), 0, 0, "cast multi transposed")


    # leading rows of the PLAIN copy scaled before the rounding (the q rows of a qkv projection, MAEST_BF16_QS); the transposed copy is not
    for t in (rnd((132, 72), 65), rnd((70, 33), 66)):
        (o, ot), = ops.cast_weights_multi([t.to(dev)], dtype, want=True, want_t=True, scaled_rows=[40], row_scale=0.1803)
        want_o = t.clone()
        want_o[:40] *= 0.1803
        close(o, want_o.to(dtype), 0, 0, "cast multi, scaled leading rows")
        close(ot, t.to(dtype).t(), 0, 0, "cast multi, transposed copy unscaled")


# --------------------------------------------------------------------------------------- LayerNorm
def case_layernorm(dev, dtype, rows):
    x = rnd((rows, 768), 7, 2.0) + 0.3
    g = 1.0 + rnd((768,), 8, 0.1)
    b = rnd((768,), 9, 0.1)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-6, dtype, save_stats=True)
    ref = F.layer_norm(x, (768,), g, b, 1e-6)
    rt, at = (1e-5, 1e-5) if dtype == torch.float32 else (1e-2, 1e-2)
    close(y, ref, rt, at, "layernorm fwd")
    close(mean, x.mean(1), 1e-5, 1e-6, "layernorm mean")
    close(rstd, 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6), 1e-5, 1e-6, "layernorm rstd")
    # residual add fused into the LayerNorm that follows it: x_new = x + delta exactly (one fp32 add per element),
    # y / statistics = those of the plain kernel on x_new, bit for bit
    delta = rnd((rows, 768), 12, 0.5).to(dtype)
    xn, y2, mean2, rstd2 = ops.add_layernorm_fwd(x.to(dev), delta.to(dev), g.to(dev), b.to(dev), 1e-6, dtype, save_stats=True)
    assert torch.equal(xn.cpu(), x + delta.float()), "fused residual add must be the exact fp32 sum"
    y3, mean3, rstd3 = ops.layernorm_fwd(xn, g.to(dev), b.to(dev), 1e-6, dtype, save_stats=True)
    assert torch.equal(y2, y3) and torch.equal(mean2, mean3) and torch.equal(rstd2, rstd3)
    if dtype == torch.float32:
        # MAEST_SPLIT3_A output (the A operand of the split-bf16 product run as one bf16 GEMM over 3 K): rows [ hi | hi | lo ] with
        # hi = bf16(y), lo = bf16(y - hi) of the fp32 result, from both kernels
        s3 = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-6, ops.SPLIT3).cpu()
        yf = y.cpu()
        hi = yf.bfloat16()
        lo = (yf - hi.float()).bfloat16()
        assert s3.shape == (rows, 2304) and s3.dtype == torch.bfloat16
        assert torch.equal(s3[:, :768], hi) and torch.equal(s3[:, 768:1536], hi) and torch.equal(s3[:, 1536:], lo), "layernorm split3 rows"
        xn3, s3b = ops.add_layernorm_fwd(x.to(dev), delta.to(dev), g.to(dev), b.to(dev), 1e-6, ops.SPLIT3)
        y2f = y2.cpu()
        hi2 = y2f.bfloat16()
        assert torch.equal(xn3, xn) and torch.equal(s3b.cpu(), torch.cat([hi2, hi2, (y2f - hi2.float()).bfloat16()], 1)), "add + layernorm split3 rows"
    # backward
    dy = rnd((rows, 768), 10).to(dtype)
    dres = rnd((rows, 768), 11)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.layer_norm(xr, (768,), gr, br, 1e-6).backward(dy.float())
    dg = torch.zeros(768, device=dev)
    db = torch.zeros(768, device=dev)
    dx, dx_lp = ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dres.to(dev), dg, db, lp_dtype=dtype)
    close(dx, xr.grad + dres, 1e-4, 1e-5, "layernorm dx")
    close(dx_lp, xr.grad + dres, *( (1e-4, 1e-5) if dtype == torch.float32 else (1e-2, 1e-2)), "layernorm dx_lp")
    close(dg, gr.grad, 1e-4, 1e-4 * math.sqrt(rows), "layernorm dgamma")
    close(db, br.grad, 1e-4, 1e-4 * math.sqrt(rows), "layernorm dbeta")
    # compact residual gradient (the first 2 tokens of every clip of n_tok tokens; zero for the others): bit for bit
    # the dense call on the scattered tensor
    for n_tok in (1, 3, 11):
        if rows % n_tok or n_tok < 2 and rows < 2:
            continue
        n_head = min(2, n_tok)
        clips = rows // n_tok
        dres_c = rnd((clips * n_head, 768), 13)
        dense = torch.zeros(clips, n_tok, 768)
        dense[:, :n_head] = dres_c.reshape(clips, n_head, 768)
        dg1, db1 = torch.zeros(768, device=dev), torch.zeros(768, device=dev)
        want, want_lp = ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dense.reshape(rows, 768).to(dev),
                                          dg1, db1, lp_dtype=dtype)
        dg2, db2 = torch.zeros(768, device=dev), torch.zeros(768, device=dev)
        got, got_lp = ops.layernorm_bwd(dy.to(dev), x.to(dev), g.to(dev), mean, rstd, dres_c.to(dev), dg2, db2,
                                        lp_dtype=dtype, head_tokens=(n_tok, n_head))
        assert torch.equal(got, want) and torch.equal(got_lp, want_lp), f"compact dres, n_tok={n_tok}"


# --------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, B, N, scale):
    q, k, v = qkv.reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
    att = ((q @ k.transpose(-2, -1)) * scale).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(B * N, 768), torch.logsumexp((q @ k.transpose(-2, -1)) * scale, -1)


def case_attention(dev, dtype, B, N, seed=20, spike=False, bf16_tol=3e-2, fwd_tol=2e-2, qs=False):
    """qs (bf16 only): the MAEST_BF16_QS contract -- the q columns of the tensor handed to the kernels hold q' = scale * log2(e) * q
    (rounded once); the oracle runs on q = q' / (scale * log2(e)) and dQ is compared as the gradient with respect to that q."""
    assert not qs or dtype == torch.bfloat16
    attn_fwd, attn_bwd = partial(ops.attn_fwd, q_prescaled=qs), partial(ops.attn_bwd, q_prescaled=qs)
    qkv = rnd((B * N, 2304), seed, 1.0).to(dtype)
    if spike:  # force a large running-max jump at a late key tile (online-softmax rescale branch)
        qf = qkv.float().clone()
        key = min(N - 1, 70)
        qf[key, 768:768 + 64] = qf[3, 0:64] * 6.0
        qkv = qf.to(dtype)
    scale = 0.125
    x = qkv.float()
    if qs:
        c = scale * 1.4426950408889634
        qp = (qkv[:, :768].float() * c).to(dtype)          # what the row-scaled projection writes
        qkv = torch.cat([qp, qkv[:, 768:]], dim=1).contiguous()
        x = torch.cat([qp.float() / c, x[:, 768:]], dim=1)
    out, lse = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
    x = x.requires_grad_(True)
    ref, ref_lse = _attn_ref(x, B, N, scale)
    rt, at = (2e-5, 2e-5) if dtype == torch.float32 else (fwd_tol, fwd_tol)
    close(out, ref, rt, at, "attention fwd")
    close(lse, ref_lse, 1e-4, 1e-4 if dtype == torch.float32 else fwd_tol, "attention lse")
    if dtype == torch.bfloat16:
        # the call above took the shape's default: the persistent one-wave-per-SIMD kernel for N > 320, four-wave workgroups with
        # LDS-DMA-fed tiles below.  Every form against the oracle: the four-wave DMA form (2), the register-staged form every other
        # dtype uses (1) -- those two bit for bit (the same products in the same order; only the tile staging differs) -- and the
        # persistent form (3) at this N whatever it is (its Q is pre-scaled by scale * log2 e and rounded to bf16 once more, its
        # row sums are those of the rounded probabilities: close, not equal)
        with ops.options(attn_fwd=2):
            out2, lse2 = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
        close(out2, ref, rt, at, "attention fwd (four-wave workgroups, LDS-DMA tiles)")
        with ops.options(attn_fwd=1):
            out1, lse1 = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
        close(out1, ref, rt, at, "attention fwd (register-staged tiles)")
        assert torch.equal(out1, out2) and torch.equal(lse1, lse2), "DMA-fed and register-staged attention forward differ"
        with ops.options(attn_fwd=3):
            out3, lse3 = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
            out3b, lse3b = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
        close(out3, ref, rt, at, "attention fwd (persistent)")
        close(lse3, ref_lse, 1e-4, fwd_tol, "attention lse (persistent)")
        assert torch.equal(out3, out3b) and torch.equal(lse3, lse3b), "the persistent attention forward does not repeat bit for bit"
        # other workgroup sizes of the four-wave form: the same per-wave arithmetic, other tile dealing (GPU only: the host emulator
        # takes seconds per launch and the CPU suite has to stay short)
        for nw in (() if _lib.host_emulation() else (5, 6, 8)):
            with ops.options(attn_fwd=2, attn_fwd_waves=nw):
                outw, lsew = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
            assert torch.equal(outw, out2) and torch.equal(lsew, lse2), f"attention forward with {nw} waves per workgroup differs"
    # backward (the oracle's autograd on the same rounded operands)
    dout = rnd((B * N, 768), seed + 1).to(dtype)
    ref.backward(dout.float())
    out_ref_lp = ref.detach().to(dtype)
    dqkv = attn_bwd(qkv.to(dev), out_ref_lp.to(dev), dout.to(dev), ref_lse.detach().contiguous().to(dev), B, N, scale)
    rt, at = (1e-4, 1e-4) if dtype == torch.float32 else (bf16_tol, bf16_tol)
    g = x.grad
    close(dqkv[:, 1536:], g[:, 1536:], rt, at, "attention dV")
    close(dqkv[:, 768:1536], g[:, 768:1536], rt, at, "attention dK")
    close(dqkv[:, :768], g[:, :768], rt, at, "attention dQ")
    if dtype == torch.bfloat16:
        # the two-kernel dK/dV + dQ form streams its tiles by LDS-DMA (unpadded, swizzled); the register-staged padded tiles
        # (attn_bwd = 4) run the same products in the same order: bit for bit, ragged last tiles included
        args = (qkv.to(dev), out_ref_lp.to(dev), dout.to(dev), ref_lse.detach().contiguous().to(dev), B, N, scale)
        with ops.options(attn_bwd=1):
            dq_dma = attn_bwd(*args)
        with ops.options(attn_bwd=4):
            dq_reg = attn_bwd(*args)
        close(dq_dma, g, rt, at, "attention backward (two-kernel, DMA-fed tiles)")
        assert torch.equal(dq_dma, dq_reg), "DMA-fed and register-staged two-kernel attention backward differ"
    if dtype == torch.bfloat16:
        # the pairing the model runs: the backward fed with the out / lse its OWN forward wrote (above: the oracle's), for the
        # four-wave form and for the persistent form -- whose lse is the log-sum of its rounded probabilities, so that the P the
        # backward recomputes does not sum to exactly 1 per row: inside the same tolerance against autograd
        for form, what in ((2, "four-wave"), (3, "persistent")):
            with ops.options(attn_fwd=form):
                o_f, lse_f = attn_fwd(qkv.to(dev), B, N, scale, save_lse=True)
            dq_f = attn_bwd(qkv.to(dev), o_f, dout.to(dev), lse_f, B, N, scale)
            # (raw q through the persistent forward: that kernel rounds scale * log2(e) * q to bf16 a second time, the backward does
            # not -- on a forced spike the two disagree by ~1.5e-2 on the dominating exponent; the model feeds pre-scaled q, `qs`)
            k = 2.0 if (form == 3 and spike and not qs) else 1.0
            close(dq_f, g, k * rt, k * at, f"attention backward on the {what} forward's own out / lse")
    if dtype == torch.bfloat16 and N <= 320:
        # the call above took the fused one-pass kernel (bf16, <= 10 key blocks); the two-kernel dK/dV + dQ form must
        # agree with the oracle too, and the two with each other to bf16 rounding of the same quantities
        with ops.options(attn_bwd=1):
            dq2 = attn_bwd(qkv.to(dev), out_ref_lp.to(dev), dout.to(dev), ref_lse.detach().contiguous().to(dev), B, N, scale)
        close(dq2[:, 1536:], g[:, 1536:], rt, at, "attention dV (two-kernel)")
        close(dq2[:, 768:1536], g[:, 768:1536], rt, at, "attention dK (two-kernel)")
        close(dq2[:, :768], g[:, :768], rt, at, "attention dQ (two-kernel)")
        close(dqkv, dq2.float(), 2e-2, 2e-2, "fused vs two-kernel attention backward")
        if N > 256:
            # the default call above took the PERSISTENT form (one workgroup per CU walking its (batch, head) items); the
            # one-workgroup-per-item form against the oracle as well, and the two bit for bit (same sums in the same order)
            with ops.options(attn_bwd=3):
                dq4 = attn_bwd(qkv.to(dev), out_ref_lp.to(dev), dout.to(dev), ref_lse.detach().contiguous().to(dev), B, N, scale)
            close(dq4, g, rt, at, "attention backward (fused, one workgroup per item)")
            assert torch.equal(dq4, dqkv), "persistent and per-item fused attention backward differ"


def case_attention_head_rows(dev, dtype, B, N, seed=25):
    """The last block's attention: only the first two queries of every clip are wanted.  Forward: the rows the
    restricted kernel writes (the 32-row tile holding them) equal the complete kernel's bit for bit.  Backward (fused bf16
    kernel): equal -- to bf16 rounding of the same sums -- to the complete backward fed a dO that is zero beyond row 2, with
    dQ = 0 for every other query; shapes the fused kernel does not serve are refused loudly."""
    from maest_amd._lib import MaestHipError
    qkv = rnd((B * N, 2304), seed, 1.0).to(dtype).to(dev)
    scale = 0.125
    # (the complete pass in the four-wave form the restricted pass is a subset of: above 320 tokens the default complete pass
    # is the persistent kernel, equal to rounding only -- checked next)
    with ops.options(attn_fwd=2 if dtype == torch.bfloat16 else 0):
        full, lse_full = ops.attn_fwd(qkv, B, N, scale, save_lse=True)
    part, lse_part = ops.attn_fwd(qkv, B, N, scale, save_lse=True, q_rows=2)
    nv = min(32, N)
    f3, p3 = full.reshape(B, N, 768), part.reshape(B, N, 768)
    assert torch.equal(p3[:, :nv], f3[:, :nv]), "restricted forward differs on the rows it computes"
    assert torch.equal(lse_part[:, :, :nv], lse_full[:, :, :nv])
    dflt, lse_dflt = ops.attn_fwd(qkv, B, N, scale, save_lse=True)
    close(dflt.float().cpu(), full.float().cpu(), 2e-2, 2e-2, "default complete forward vs four-wave form")
    close(lse_dflt.cpu(), lse_full.cpu(), 1e-4, 2e-2, "default complete forward vs four-wave form (lse)")
    # gather / scatter of the head tokens' rows
    comp = ops.gather_head_rows(full, B, N, 2)
    assert torch.equal(comp.reshape(B, 2, 768), f3[:, :2])
    back = ops.scatter_head_rows(comp, B, N, 2, nv).reshape(B, N, 768)
    assert torch.equal(back[:, :2], f3[:, :2]) and not back[:, 2:nv].any()
    xf = rnd((B * N, 768), seed + 3).to(dev)
    assert torch.equal(ops.gather_head_rows(xf, B, N, 2).reshape(B, 2, 768), xf.reshape(B, N, 768)[:, :2])
    dout_c = rnd((B * 2, 768), seed + 1).to(dtype).to(dev)
    dense = ops.scatter_head_rows(dout_c, B, N, 2, N)          # zero everywhere else
    if ops.attn_bwd_rows_supported(dtype, N):
        want = ops.attn_bwd(qkv, full, dense, lse_full, B, N, scale)
        got = ops.attn_bwd(qkv, part, ops.scatter_head_rows(dout_c, B, N, 2, nv), lse_part, B, N, scale, q_rows=2)
        close(got[:, 768:], want[:, 768:].float().cpu(), 2e-2, 2e-2, "restricted attention backward dK, dV")
        g3, w3 = got.reshape(B, N, 2304), want.reshape(B, N, 2304)
        close(g3[:, :2, :768], w3[:, :2, :768].float().cpu(), 2e-2, 2e-2, "restricted attention backward dQ")
        assert not g3[:, 2:, :768].any(), "queries without gradient must get dQ = 0"
        with ops.options(attn_bwd=1):
            two = ops.attn_bwd(qkv, full, dense, lse_full, B, N, scale)
        close(got[:, 768:], two[:, 768:].float().cpu(), 2e-2, 2e-2, "restricted fused vs complete two-kernel backward")
    else:
        try:
            ops.attn_bwd(qkv, full, dense, lse_full, B, N, scale, q_rows=2)
        except MaestHipError as e:
            assert "q_rows" in str(e)
        else:
            raise AssertionError("restricted backward must be refused on shapes the fused kernel does not serve")


# ----------------------------------------------------------------------------- patch embed pieces
def case_patch_embed(dev, dtype, B, T, patchout=0, mix=False, seed=30, masked=False, stride=(10, 10)):
    """stride: (frequency, time) step of the 16 x 16 patches (models/maest.py:214-241; every published architecture: (10, 10))."""
    Fdim = 96
    x = rnd((B, Fdim, T), seed)
    Tp = (T - 16) // stride[1] + 1
    Fp = (Fdim - 16) // stride[0] + 1
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    keep = np.sort(rng.permutation(Tp)[: Tp - patchout]).astype(np.int32) if patchout else None
    Tk = Tp - patchout
    t_list = torch.arange(Tp) if keep is None else torch.from_numpy(keep).long()
    tok = torch.stack(torch.meshgrid(torch.arange(Fp), t_list, indexing="ij"), -1).reshape(-1, 2).to(torch.int32)
    tok_dev = tok.contiguous().to(dev)
    perm = lam = t_str = f_str = None
    xm = x
    if masked:
        # SpecMasking stripes per clip (helpers/spec_masking.py:27-33), applied by the loader BEFORE mixup; explicit
        # (start, width) lists incl. zero-width, edge-touching and overlapping stripes
        n_t, n_f = 5, 3
        t_str = torch.from_numpy(np.stack([rng.integers(0, T - 8, (B, n_t)), rng.integers(0, 9, (B, n_t))], -1).astype(np.int32))
        f_str = torch.from_numpy(np.stack([rng.integers(0, Fdim - 5, (B, n_f)), rng.integers(0, 6, (B, n_f))], -1).astype(np.int32))
        t_str[0, 0] = torch.tensor([T - 3, 8])          # runs past the right edge: clamped
        f_str[0, 0] = torch.tensor([Fdim - 2, 5])
        xm = torch.stack([O.spec_masking(x[b], [tuple(v) for v in t_str[b].tolist()], [tuple(v) for v in f_str[b].tolist()])
                          for b in range(B)])
    if mix:
        perm = torch.from_numpy(rng.permutation(B).astype(np.int32))
        lam = torch.from_numpy(rng.random(B).astype(np.float32))
        xm = O.mixup(xm, perm.long(), lam)
    cols = ops.patch_im2col(x.to(dev), tok_dev, dtype, stride=stride,
                            perm=None if perm is None else perm.to(dev), lam=None if lam is None else lam.to(dev),
                            t_stripes=None if t_str is None else t_str.to(dev), f_stripes=None if f_str is None else f_str.to(dev))
    if masked and not mix and dtype == torch.float32:
        # the fused predicate must equal the stand-alone kernel (maest_spec_mask) bit for bit
        xs = ops.spec_mask_(x.clone().to(dev), t_str.to(dev), f_str.to(dev))
        cols2 = ops.patch_im2col(xs, tok_dev, dtype, stride=stride)
        assert torch.equal(cols, cols2), "fused SpecMasking differs from spec_mask_ + im2col"
    # a float16 batch (what the reference's loader hands out, discogs/dataset.py:58-67) widened inside the load must
    # equal the same values passed as fp32, bit for bit
    xh = x.half()
    kw = dict(stride=stride, perm=None if perm is None else perm.to(dev), lam=None if lam is None else lam.to(dev),
              t_stripes=None if t_str is None else t_str.to(dev), f_stripes=None if f_str is None else f_str.to(dev))
    assert torch.equal(ops.patch_im2col(xh.to(dev), tok_dev, dtype, **kw),
                       ops.patch_im2col(xh.float().to(dev), tok_dev, dtype, **kw)), "fp16 input path differs from x.float()"
    ref = F.unfold(xm.unsqueeze(1), kernel_size=16, stride=stride)      # [B, 256, Fp*Tp]
    ref = ref.reshape(B, 256, Fp, Tp)
    if keep is not None:
        ref = ref[:, :, :, torch.from_numpy(keep).long()]
    ref = ref.permute(0, 2, 3, 1).reshape(B * Fp * Tk, 256)
    close(cols, ref.to(dtype), 0, 1e-6 if dtype == torch.float32 else 0, "im2col")
    # token assembly vs oracle.tokens_from_patches
    Tt = (62 if T <= 640 else T // 10) if stride[1] == 10 else Tp + 1
    sd = {"cls_token": rnd((1, 1, 768), 40, .02), "dist_token": rnd((1, 1, 768), 41, .02),
          "new_pos_embed": rnd((1, 2, 768), 42, .02), "freq_new_pos_embed": rnd((1, 768, Fp, 1), 43, .02),
          "time_new_pos_embed": rnd((1, 768, 1, Tt), 44, .02)}
    toff = 0 if patchout == 0 else min(1, Tt - Tp)
    conv = rnd((B, 768, Fp, Tp), 45)
    want = O.tokens_from_patches(conv, sd, toffset=toff, t_keep=None if keep is None else keep.tolist())
    convk = conv if keep is None else conv[:, :, :, torch.from_numpy(keep).long()]
    patches = convk.permute(0, 2, 3, 1).reshape(B * Fp * Tk, 768).contiguous()
    x0 = ops.token_assemble(patches.to(dev), sd["cls_token"].reshape(768).to(dev), sd["dist_token"].reshape(768).to(dev),
                            sd["new_pos_embed"].reshape(2, 768).contiguous().to(dev),
                            sd["freq_new_pos_embed"].reshape(768, Fp).contiguous().to(dev),
                            sd["time_new_pos_embed"].reshape(768, Tt).contiguous().to(dev), toff, tok_dev, B)
    close(x0, want, 0, 1e-6, "token assemble")
    # backward of token assembly
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    convg = conv.clone().requires_grad_(True)
    dx0 = rnd(tuple(want.shape), 46)
    O.tokens_from_patches(convg, sdg, toffset=toff, t_keep=None if keep is None else keep.tolist()).backward(dx0)
    z = lambda *s: torch.zeros(*s, device=dev)
    d_cls, d_dist, d_np, d_fp, d_tp = z(768), z(768), z(2, 768), z(768, Fp), z(768, Tt)
    dp = ops.token_assemble_bwd(dx0.to(dev), B, Fp, Tt, toff, tok_dev, dtype, d_cls, d_dist, d_np, d_fp, d_tp)
    gk = convg.grad if keep is None else convg.grad[:, :, :, torch.from_numpy(keep).long()]
    close(dp, gk.permute(0, 2, 3, 1).reshape(B * Fp * Tk, 768), *((0, 1e-6) if dtype == torch.float32 else (1e-2, 1e-2)), "dpatches")
    close(d_cls, sdg["cls_token"].grad.reshape(768), 1e-5, 1e-5, "d cls")
    close(d_dist, sdg["dist_token"].grad.reshape(768), 1e-5, 1e-5, "d dist")
    close(d_np, sdg["new_pos_embed"].grad.reshape(2, 768), 1e-5, 1e-5, "d new_pos")
    close(d_fp, sdg["freq_new_pos_embed"].grad.reshape(768, Fp), 1e-4, 1e-4, "d freq_pos")
    close(d_tp, sdg["time_new_pos_embed"].grad.reshape(768, Tt), 1e-4, 1e-4, "d time_pos")


# ------------------------------------------------------------------------------------------- head
def case_head(dev, B, N):
    x = rnd((B, N, 768), 50, 1.5)
    g = 1.0 + rnd((768,), 51, 0.1)
    b = rnd((768,), 52, 0.1)
    cls, dist, feat, mean, rstd = ops.head_pool_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-6, save_stats=True)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    xn = F.layer_norm(xr, (768,), gr, br, 1e-6)
    close(cls, xn[:, 0], 1e-5, 1e-5, "head cls")
    close(dist, xn[:, 1], 1e-5, 1e-5, "head dist")
    close(feat, (xn[:, 0] + xn[:, 1]) / 2, 1e-5, 1e-5, "head feat")
    dc, dd, df = rnd((B, 768), 53), rnd((B, 768), 54), rnd((B, 768), 55)
    ((xn[:, 0] * dc).sum() + (xn[:, 1] * dd).sum() + (((xn[:, 0] + xn[:, 1]) / 2) * df).sum()).backward()
    dg = torch.zeros(768, device=dev)
    db = torch.zeros(768, device=dev)
    dx = ops.head_pool_bwd(dc.to(dev), dd.to(dev), df.to(dev), x.to(dev), g.to(dev), mean, rstd, dg, db)
    close(dx, xr.grad, 1e-4, 1e-5, "head dx")
    close(dg, gr.grad, 1e-4, 1e-4, "head dgamma")
    close(db, br.grad, 1e-4, 1e-4, "head dbeta")
    emb = ops.embed_pool(x.to(dev))
    close(emb, torch.cat([x[:, 0], x[:, 1], x[:, 2:].mean(1)], 1), 1e-5, 1e-5, "embed pool")


def case_loss(dev, rows, cols):
    z = rnd((rows, cols), 60, 2.0)
    rng = np.random.Generator(np.random.PCG64(61))
    y = torch.from_numpy((rng.random((rows, cols)) < 0.1).astype(np.float32))
    perm = torch.from_numpy(rng.permutation(rows).astype(np.int32))
    lam = torch.from_numpy(rng.random(rows).astype(np.float32))
    zr = z.clone().requires_grad_(True)
    ym = O.mixup(y, perm.long(), lam)
    ref = F.binary_cross_entropy_with_logits(zr, ym)
    ref.backward()
    loss, dz = ops.bce_logits(z.to(dev), y.to(dev), 1.0, perm.to(dev), lam.to(dev))
    close(loss, ref.detach(), 1e-5, 1e-6, "bce loss")
    close(dz, zr.grad, 1e-4, 1e-8, "bce dlogits")
    loss2, _ = ops.bce_logits(z.to(dev), y.to(dev), 0.5)
    close(loss2, 0.5 * F.binary_cross_entropy_with_logits(z, y), 1e-5, 1e-6, "bce weighted")
    act = ops.sigmoid_mean(z.to(dev))
    close(act, torch.sigmoid(z).mean(0), 1e-5, 1e-6, "sigmoid mean")
    src = rnd((rows, cols), 62)
    out = torch.zeros(cols, device=dev)
    ops.colsum(src.to(dev), out)
    close(out, src.sum(0), 1e-4, 1e-4, "colsum")
    srcb = src.to(torch.bfloat16)
    out = torch.zeros(cols, device=dev)
    ops.colsum(srcb.to(dev), out)
    close(out, srcb.float().sum(0), 1e-4, 1e-4, "colsum bf16")
    v = rnd((1000,), 63)
    w = ops.scale_(v.clone().to(dev), 0.25)
    close(w, v * 0.25, 0, 0, "scale")


def case_spec_mask(dev, B, T):
    x = rnd((B, 96, T), 70)
    rng = np.random.Generator(np.random.PCG64(71))
    ts = np.stack([rng.integers(0, T - 8, (B, 5)), rng.integers(0, 8, (B, 5))], -1).astype(np.int32)
    fs = np.stack([rng.integers(0, 96 - 5, (B, 3)), rng.integers(0, 5, (B, 3))], -1).astype(np.int32)
    want = torch.stack([O.spec_masking(x[b], [tuple(p) for p in ts[b]], [tuple(p) for p in fs[b]]) for b in range(B)])
    got = ops.spec_mask_(x.clone().to(dev), torch.from_numpy(ts).to(dev), torch.from_numpy(fs).to(dev))
    close(got, want, 0, 0, "spec mask")


def case_mel(dev, B, S, seed=80):
    from maest_amd.melspectrogram import MelSpectrogram
    rng = np.random.Generator(np.random.PCG64(seed))
    w = torch.from_numpy((rng.random((B, S), dtype=np.float32) * 2 - 1) * 0.5)
    mel = MelSpectrogram()
    got = mel(w.to(dev))
    want = O.logmel(w)
    assert got.shape == want.shape == (B, 96, 1 + S // 256)
    # north_star tolerance for floating point: 1e-3 relative (values are O(1) after log compression)
    close(got, want, 1e-3, 1e-3, "logmel")
    err = (got.cpu() - want).abs().max().item()
    assert err < 2e-4, f"logmel max abs err {err}"


# ------------------------------------------------------------------ on-disk mel chunks -> input
def case_melfile(dev, tmp_path, size=50, counts=(80, 30, 31, 50, 1, 45), offsets=(13, 0, 0, 0, 0, 20), seed=90):
    """MelFileReader (host plan + one device kernel) against the oracle restatement of the reference reader,
    bit-exact (float16 arithmetic reproduced on the device): plain slice, short files (even / odd padding),
    exact length, a single frame, and an offset that runs past the end of the file."""
    from maest_amd.melfile import MelFileReader
    from oracle import melfile_oracle as MO
    rd = MelFileReader(tmp_path, clip_length=1, sample_rate=size, hop_size=1)     # melspectrogram_size = size
    assert rd.melspectrogram_size == size
    rng = np.random.Generator(np.random.PCG64(seed))
    names = []
    for i, n in enumerate(counts):
        fr = (rng.random((n, 96), dtype=np.float32) * 5.0).astype("float16")
        name = f"clip{i}.mel"
        fr.tofile(tmp_path / name)
        names.append(name)
    for normalize in (True, False):
        got = rd.load_batch(names, dev, offsets=list(offsets), normalize=normalize)
        assert got.shape == (len(counts), 1, 96, size) and got.dtype == torch.float32
        for i, name in enumerate(names):
            want = MO.load_melspectrogram(tmp_path / name, size, 96, offsets[i])
            if normalize:
                want = MO.norm_func(want)
            assert want.dtype == np.float16
            w32 = torch.from_numpy(want.astype(np.float32))
            assert torch.equal(got[i].cpu(), w32), (
                f"melfile clip {i} normalize={normalize}: max diff {(got[i].cpu() - w32).abs().max().item()}")


# ------------------------------------------------------------------ second mel parameterisation
def case_augment_mel(dev, B, S, seed=95):
    """AugmentMelSTFT (csrc/mel2.hip) against the oracle restatement: eval mode, and training mode with the
    band-edge jitter and stripes replayed from the same torch seed."""
    from maest_amd.preprocess import AugmentMelSTFT
    rng = np.random.Generator(np.random.PCG64(seed))
    w = torch.from_numpy((rng.random((B, S), dtype=np.float32) * 2 - 1) * 0.5)
    m = AugmentMelSTFT().to(dev).eval()
    got = m(w.to(dev))
    want = O.augment_mel(w)
    assert got.shape == want.shape == (B, 128, 1 + (S - 1) // 320)
    close(got, want, 1e-3, 1e-3, "augment mel (eval)")
    # training: the reference draws fmin, fmax (torch.randint x2), then the frequency and the time stripe (rand x2 each)
    m = AugmentMelSTFT(fmin_aug_range=10, fmax_aug_range=2000).to(dev).train()
    torch.manual_seed(7)
    got = m(w.to(dev))
    torch.manual_seed(7)
    fmin = 0.0 + torch.randint(10, (1,)).item()
    fmax = m.fmax + 2000 // 2 - torch.randint(2000, (1,)).item()
    T = want.shape[-1]
    v = torch.rand(1) * 48; mv = torch.rand(1) * (128 - v); fs = (int(mv.long()), int(v.long()))
    v = torch.rand(1) * 192; mv = torch.rand(1) * (T - v); ts = (int(mv.long()), int(v.long()))
    want = O.augment_mel(w, fmin=fmin, fmax=fmax, f_stripe=fs, t_stripe=ts)
    close(got, want, 1e-3, 1e-3, "augment mel (train)")


# ------------------------------------------------------------------ stochastic weight averaging
def case_swa(dev):
    shapes = [(768, 33), (5,), (4097,), (3, 1, 16, 16)]
    avg = [rnd(sh, 110 + i) for i, sh in enumerate(shapes)]
    want = [a.clone() for a in avg]
    got = [a.clone().to(dev) for a in avg]
    for step in range(3):
        cur = [rnd(sh, 120 + 10 * step + i) for i, sh in enumerate(shapes)]
        inv = 1.0 / (step + 2)
        ops.swa_update_multi(got, [c.to(dev) for c in cur], inv)
        want = [w + (c - w) * inv for w, c in zip(want, cur)]
    for g, w in zip(got, want):
        close(g, w, 1e-6, 1e-7, "swa running mean")


# ------------------------------------------------------------------------------------- split-bf16 ("bf16x3") products
def case_split_precision(dev, M=512, N=256, K=192, B=1, Ntok=75):
    """fp32 tensors with the matrix products as three bf16 MFMAs on hi/lo operand splits (common.h: mma_chunk2):
    the 256-tile NT GEMM (every epilogue path shares the main loop: checked on bias + residual) and the attention
    forward, against fp64 references.  Gate: 1e-4 of the output scale -- 40x tighter than plain bf16 operands manage
    (~4e-3) and an order of magnitude inside the 1e-3 parity gate; plain-bf16 results on the same data are asserted
    to be well OUTSIDE it, so the test cannot pass on a silently taken bf16 path."""
    a, b, bias, res = rnd((M, K), 50), rnd((N, K), 51), rnd((N,), 52), rnd((M, N), 53)
    ref = (a.double() @ b.double().t() + bias.double() + res.double())
    scale = ref.abs().max().item()
    with ops.options(gemm_min_m=512):
        c = ops.gemm_nt(a.to(dev), b.to(dev), bias.to(dev), out_dtype=torch.float32, epi=ops.EPI_RESIDUAL,
                        aux_in=res.to(dev), x3=True)
    qkv = rnd((B * Ntok, 2304), 54)
    out, lse = ops.attn_fwd(qkv.to(dev), B, Ntok, 0.125, save_lse=True, x3=True)
    # backward products: the wgrad form (256-tile TN kernel) and the attention backward
    ta, tb = rnd((288, 256), 55), rnd((288, 512), 56)
    tout = torch.zeros((256, 512), dtype=torch.float32, device=dev)
    tcs = torch.zeros(256, dtype=torch.float32, device=dev)
    with ops.options(gemm_variant=4):
        ops.gemm_tn(ta.to(dev), tb.to(dev), tout, colsum=tcs, split_k=0, x3=True)
    xq = qkv.double().requires_grad_(True)
    oref, lref = _attn_ref(xq, B, Ntok, 0.125)
    dout = rnd((B * Ntok, 768), 57)
    oref.backward(dout.double())
    dqkv = ops.attn_bwd(qkv.to(dev), oref.detach().float().to(dev), dout.to(dev), lref.detach().float().contiguous().to(dev),
                        B, Ntok, 0.125, x3=True)
    tref = ta.double().t() @ tb.double()
    et = (tout.double().cpu() - tref).abs().max().item() / tref.abs().max().item()
    assert et < 1e-4, f"split-bf16 TN GEMM: {et:.2e} of the output scale"
    assert (tcs.double().cpu() - ta.double().sum(0)).abs().max().item() < 1e-4
    eb = (dqkv.double().cpu() - xq.grad).abs().max().item() / xq.grad.abs().max().item()
    assert eb < 1e-4, f"split-bf16 attention backward: {eb:.2e} of the gradient scale"
    e = (c.double().cpu() - ref).abs().max().item() / scale
    assert e < 1e-4, f"split-bf16 GEMM: {e:.2e} of the output scale"
    with ops.options(gemm_min_m=512):
        c16 = ops.gemm_nt(a.bfloat16().to(dev), b.bfloat16().to(dev), bias.to(dev), out_dtype=torch.float32,
                          epi=ops.EPI_RESIDUAL, aux_in=res.to(dev))
    e16 = (c16.double().cpu() - ref).abs().max().item() / scale
    assert e16 > 10 * e, f"plain bf16 operands ({e16:.2e}) should be far coarser than the split ({e:.2e})"
    # the same three-term product as ONE bf16 GEMM over 3 K (MAEST_SPLIT3_A x MAEST_SPLIT3_B rows; the bf16 kernels, fp32 output): the weight
    # rows come from maest_cast_weights_multi, the activation rows are built here as the LayerNorm / attention kernels write them
    b3 = ops.cast_weights_multi([b.to(dev)], ops.SPLIT3)[0][0]
    bh = b.bfloat16()
    assert torch.equal(b3.cpu(), torch.cat([bh, (b - bh.float()).bfloat16(), bh], 1)), "split3 weight rows"
    ah = a.bfloat16()
    a3 = torch.cat([ah, ah, (a - ah.float()).bfloat16()], 1).contiguous()
    with ops.options(gemm_min_m=512):
        c3 = ops.gemm_nt(a3.to(dev), b3, bias.to(dev), out_dtype=torch.float32, epi=ops.EPI_RESIDUAL, aux_in=res.to(dev))
    e3 = (c3.double().cpu() - ref).abs().max().item() / scale
    assert e3 < 1e-4, f"split-bf16 GEMM as one bf16 GEMM over 3 K: {e3:.2e} of the output scale"
    # ... and the GELU epilogue writing its result in the same row form (the fc1 -> fc2 hand-over): hi / lo of the fp32-output epilogue's values
    gref = F.gelu(a.double() @ b.double().t() + bias.double())
    for big in (True, False):      # the staged epilogue form of the 256-row-tile kernel; the element-wise one of the 128 x 128 kernel (small M)
        with ops.options(gemm_min_m=512 if big else 1 << 30):
            gf = ops.gemm_nt(a3.to(dev), b3, bias.to(dev), out_dtype=torch.float32, epi=ops.EPI_GELU).cpu()
            g3 = ops.gemm_nt(a3.to(dev), b3, bias.to(dev), out_dtype=ops.SPLIT3, epi=ops.EPI_GELU).cpu()
        gh = gf.bfloat16()
        assert g3.shape == (M, 3 * N) and torch.equal(g3[:, :N], g3[:, N:2 * N]), "split3 rows: the two hi thirds"
        same_kernel = (not big) or ops.gemm_split3_out_fast(M, N, 3 * K)
        if same_kernel:      # both outputs left the same main loop: the split of the very fp32 values
            assert torch.equal(g3, torch.cat([gh, gh, (gf - gh.float()).bfloat16()], 1)), f"GELU epilogue split3 rows (big = {big})"
        # (a build without the one-wave-per-SIMD kernel writes the fp32 form from the eight-wave kernel and the split form from the 128 x 128 one)
        rec = g3[:, :N].double() + g3[:, 2 * N:].double()
        assert (rec - gref).abs().max().item() / gref.abs().max().item() < 1e-4, "hi + lo of the split rows"
        assert (gf.double() - gref).abs().max().item() / gref.abs().max().item() < 1e-4
    o3 = ops.attn_fwd(qkv.to(dev), B, Ntok, 0.125, x3=True, out_split3=True).cpu()
    of = out.cpu()
    oh = of.bfloat16()
    assert torch.equal(o3, torch.cat([oh, oh, (of - oh.float()).bfloat16()], 1)), "attention forward split3 rows"
    oref, lref = oref.detach(), lref.detach()
    eo = (out.double().cpu() - oref).abs().max().item() / oref.abs().max().item()
    el = (lse.double().cpu() - lref).abs().max().item()
    assert eo < 1e-4 and el < 1e-4, f"split-bf16 attention forward: out {eo:.2e}, lse {el:.2e}"
    # (the call above took the kernel that splits the K / V tiles once while staging them; the per-use split form behind attn_fwd = 1)
    with ops.options(attn_fwd=1):
        out1, lse1 = ops.attn_fwd(qkv.to(dev), B, Ntok, 0.125, save_lse=True, x3=True)
    eo1 = (out1.double().cpu() - oref).abs().max().item() / oref.abs().max().item()
    assert eo1 < 1e-4 and (lse1.double().cpu() - lref).abs().max().item() < 1e-4, f"split-bf16 attention forward (per-use split): {eo1:.2e}"
    return e, eo
