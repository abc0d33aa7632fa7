"""Tensor-level wrappers over the C ABI (include/maest_hip.h).  PyTorch is used here only as the
owner of device memory and of the HIP stream; every computation happens in libmaest_hip.so."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import BF16, BF16_QS, F32X3_A3, SPLIT3_A, SPLIT3_B, EPI_ATOMIC, EPI_MUL, EPI_GELU, EPI_NONE, EPI_RESIDUAL, F16, F32, F32X3, call

DT = {torch.float32: F32, torch.bfloat16: BF16}
SPLIT3 = "split3"     # output "dtype" of the LayerNorm / attention-forward wrappers and of cast_weights_multi: the split-bf16 operand rows of
                      # include/maest_hip.h (MAEST_SPLIT3_A / _B): bf16 [rows, 3 * cols], one bf16 GEMM over 3 K = the bf16x3 product

def _mm_code(dtype, x3: bool):
    """dtype code of a matrix-product operand.  `x3` ("bf16x3" precision mode): tensors stay fp32, but the product runs
    as three bf16 MFMAs on hi/lo splits of the fp32 operands (csrc/common.h: mma_chunk2) -- MAEST_F32X3.  The mode is an
    explicit argument of every matrix-product wrapper (the engine knows the model's mode); there is no process-wide
    switch, so a direct fp32 caller always gets exact fp32 products whatever other models run in the process."""
    return F32X3 if (x3 and dtype == torch.float32) else DT[dtype]


class KernelTimer:
    """HIP-event timing of C-ABI launches on the stream they are enqueued on (bench.py roofline).
    Usage: ``with ops.KernelTimer(kinds={"maest_gemm_nt"}) as t: ...; t.summary()``."""

    def __init__(self, kinds=None):
        self.kinds = kinds
        self.records = []   # (name, start_event, end_event, work)

    def __enter__(self):
        global _TIMER
        _TIMER = self
        return self

    def __exit__(self, *a):
        global _TIMER
        _TIMER = None

    def summary(self):
        """-> {name: {"launches": n, "ms": total, "work": total algorithmic flops}} (syncs)."""
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, work in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out


_TIMER = None


def _timed_call(name, work, *args, _entry=None):
    """`name`: timing bucket (KernelTimer); `_entry`: C-ABI symbol when it differs from the bucket."""
    t = _TIMER
    if t is None or (t.kinds is not None and name not in t.kinds):
        return call(_entry or name, *args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    call(_entry or name, *args)
    e1.record()
    t.records.append((name, e0, e1, work))
HEADS = 12
HEAD_DIM = 64
EMBED = 768


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda or _lib.host_emulation()):
            raise _lib.MaestHipError(
                f"maest_amd kernels need tensors on a HIP device, got {t.device}; there is no CPU fallback")
        if not t.is_contiguous():
            raise _lib.MaestHipError("maest_amd kernels need contiguous tensors")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s(t):
    if _lib.host_emulation():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------ GEMM
def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            out: Optional[torch.Tensor] = None, out_dtype=None, epi: int = EPI_NONE,
            aux_in: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None,
            split_k: int = 1, M: Optional[int] = None, N: Optional[int] = None,
            K: Optional[int] = None, x3: bool = False, split3: bool = False) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ b[N,K]^T + bias).  a/b may carry padded leading dims (2-D views of
    bigger buffers): lda/ldb are taken from stride(0).  split3: a / b are MAEST_SPLIT3_A / _B rows (K = 3 x the logical depth;
    only the timing bucket's flop count cares: 2 M N K / 3, the algorithmic work of the split product)."""
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype
    assert a.stride(1) == 1 and b.stride(1) == 1
    M = a.shape[0] if M is None else M
    N = b.shape[0] if N is None else N
    K = a.shape[1] if K is None else K
    s3out = out_dtype == SPLIT3       # gelu(acc + bias) as MAEST_SPLIT3_A rows, bf16 [M, 3 N] (bf16 operands, GELU epilogue without side output)
    if out is None:
        out = (torch.empty((M, 3 * N), dtype=torch.bfloat16, device=a.device) if s3out
               else torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device))
    assert out.stride(1) == 1
    ld_aux = 0
    for x in (aux_in, aux_out):
        if x is not None:
            assert x.stride(1) == 1
            ld_aux = x.stride(0)
    for t in (a, b, out, aux_in, aux_out):
        if t is not None and not (t.is_cuda or _lib.host_emulation()):
            raise _lib.MaestHipError("maest_amd kernels need tensors on a HIP device; there is no CPU fallback")
    _chk(bias)
    # timing bucket: the token-major GEMMs of the blocks apart from the few small ones (head, patch-embed remainder, the
    # last block's head-token rows), which run the 128x128 kernel and would blur the dominant kernel's figures
    _timed_call("maest_gemm_nt" if M >= 4096 else "maest_gemm_nt_small", 2.0 * M * N * K / (3 if split3 else 1), _p(a), a.stride(0), _p(b),
                b.stride(0), _mm_code(a.dtype, x3), _p(out), out.stride(0), SPLIT3_A if s3out else DT[out.dtype], M, N, K, _p(bias), epi,
                _p(aux_in), _p(aux_out), ld_aux, split_k, _s(a), _entry="maest_gemm_nt")
    return out


def gemm_split3_out_fast(M: int, N: int, K: int) -> bool:
    """Whether maest_gemm_nt writes out_dtype = SPLIT3 from the staged epilogue of the 256-row-tile one-wave-per-SIMD kernel (bf16 [M, K] x
    [N, K], GELU); other shapes / builds take the element-wise epilogue of the 128 x 128 kernel -- correct, but slower than an fp32 output
    (measured at M = 4480: the engine then keeps fc2 on the per-chunk split kernel)."""
    return (M >= max(512, get_option("gemm_min_m")) and N % 256 == 0 and K % 64 == 0 and get_option("gemm_variant") == 0
            and bool(_lib.kernel_forms() & _lib.FORM_GEMM_NT_OW))


def gemm_nt_rowdot(a: torch.Tensor, b: torch.Tensor, other: torch.Tensor, rows_per_item: int, *, out_dtype=None,
                   bias: Optional[torch.Tensor] = None, x3: bool = False):
    """c[M,N] = a[M,K] @ b[N,K]^T (+ bias) in out_dtype and, from the same C-tile pass, rowdot[M / rows_per_item, N / 64,
    rows_per_item] = per (row, 64-column group) dot products of the stored c with `other` (same shape / dtype as c):
    the attention backward's delta = rowsum(dO * O) out of the GEMM that produces dO (include/maest_hip.h)."""
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    out_dtype = out_dtype or a.dtype
    assert other.shape == (M, N) and other.dtype == out_dtype and other.stride(1) == 1 and M % rows_per_item == 0
    _chk(bias)
    for t in (a, b, other):
        if not (t.is_cuda or _lib.host_emulation()):
            raise _lib.MaestHipError("maest_amd kernels need tensors on a HIP device; there is no CPU fallback")
    c = torch.empty((M, N), dtype=out_dtype, device=a.device)
    rowdot = torch.empty((M // rows_per_item, N // 64, rows_per_item), dtype=torch.float32, device=a.device)
    _timed_call("maest_gemm_nt" if M >= 4096 else "maest_gemm_nt_small", 2.0 * M * N * K, _p(a), a.stride(0), _p(b),
                b.stride(0), _mm_code(a.dtype, x3), _p(c), c.stride(0), DT[out_dtype], M, N, K, _p(bias), _p(other),
                other.stride(0), _p(rowdot), rows_per_item, _s(a), _entry="maest_gemm_nt_rowdot")
    return c, rowdot


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, colsum: Optional[torch.Tensor] = None,
            split_k: int = 1, M: Optional[int] = None, N: Optional[int] = None, x3: bool = False) -> torch.Tensor:
    """out[M,N] (fp32, pre-zeroed) += a[K,M]^T @ b[K,N];  colsum[M] (fp32, pre-zeroed) += a.sum(0)."""
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype and a.shape[0] == b.shape[0]
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.dtype == torch.float32
    K = a.shape[0]
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    out2 = out.view(M, N)
    for t in (a, b, out, colsum):
        if t is not None and not (t.is_cuda or _lib.host_emulation()):
            raise _lib.MaestHipError("maest_amd kernels need tensors on a HIP device; there is no CPU fallback")
    # split-K partials through a scratch buffer where the kernel has that form (deterministic, no atomics on `out`): the buffer
    # comes from torch's caching allocator on the CURRENT stream -- the stream the call runs on -- so its reuse is stream-ordered
    code = _mm_code(a.dtype, x3)
    nbytes = ctypes.c_int64(0)
    if get_option("tn_reduce") != 0:         # (the default combines split-K partials with atomics: no scratch, no query)
        call("maest_gemm_tn_workspace_bytes", code, M, N, K, split_k, ctypes.byref(nbytes))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=a.device) if nbytes.value > 0 else None
    _timed_call("maest_gemm_tn", 2.0 * M * N * K, _p(a), a.stride(0), _p(b), b.stride(0), code, _p(out2),
                out2.stride(0), M, N, K, _p(colsum), split_k, _p(ws), nbytes.value, _s(a), _entry="maest_gemm_tn_ws")
    return out


def gemm_tn_workspace_bytes(dtype, M: int, N: int, K: int, split_k: int = 0, x3: bool = False) -> int:
    """Scratch bytes with which gemm_tn combines its split-K partials without atomics (0: this shape has no such form)."""
    nbytes = ctypes.c_int64(0)
    call("maest_gemm_tn_workspace_bytes", _mm_code(dtype, x3), M, N, K, split_k, ctypes.byref(nbytes))
    return nbytes.value


def transpose(src: torch.Tensor, ld_dst: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[c, r] = src[r, c]; out has shape [cols, ld_dst] with the pad columns zeroed."""
    assert src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    ld_dst = ld_dst or rows
    if out is None:
        out = torch.empty((cols, ld_dst), dtype=src.dtype, device=src.device)
    _chk(out)
    _timed_call("maest_transpose", 0.0, _p(src), src.stride(0), _p(out), ld_dst, rows, cols, DT[src.dtype], _s(src))
    return out


def cast_weights(src: torch.Tensor, dtype, want=True, want_t=False):
    """fp32 parameter -> (operand copy [rows, cols], transposed operand copy [cols, rows])."""
    _chk(src)
    assert src.dtype == torch.float32
    w2 = src.reshape(src.shape[0], -1)
    rows, cols = w2.shape
    dst = torch.empty((rows, cols), dtype=dtype, device=src.device) if want else None
    dst_t = torch.empty((cols, rows), dtype=dtype, device=src.device) if want_t else None
    _timed_call("maest_cast_weights", 0.0, _p(w2), _p(dst), _p(dst_t), rows, cols, DT[dtype], _s(src))
    return dst, dst_t


def cast_weights_multi(srcs, dtype, want=True, want_t=False, scaled_rows=None, row_scale=1.0):
    """Many fp32 parameters -> operand copies in ONE launch.  Returns a list of (copy | None, transposed | None).
    scaled_rows[i] > 0: the first scaled_rows[i] rows of the i-th plain copy (not of its transpose) are multiplied by row_scale before
    the rounding (the q rows of a qkv projection for MAEST_BF16_QS attention)."""
    _chk(*srcs)
    n = len(srcs)
    w2 = [s.reshape(s.shape[0], -1) for s in srcs]
    dev = srcs[0].device
    outs = []
    s3 = dtype == SPLIT3              # rows of [ hi | lo | hi ] bf16 thirds (MAEST_SPLIT3_B); no transposed copy
    assert not (s3 and want_t)
    for w in w2:
        assert w.dtype == torch.float32 and w.is_contiguous()
        r, c = w.shape
        outs.append((torch.empty((r, 3 * c if s3 else c), dtype=torch.bfloat16 if s3 else dtype, device=dev) if want else None,
                     torch.empty((c, r), dtype=dtype, device=dev) if want_t else None))
    vp = ctypes.c_void_p * n
    ip = ctypes.c_int * n
    ptr = lambda t: 0 if t is None else t.data_ptr()
    a_src = vp(*[w.data_ptr() for w in w2])
    a_dst = vp(*[ptr(o[0]) for o in outs])
    a_dt = vp(*[ptr(o[1]) for o in outs])
    a_r = ip(*[w.shape[0] for w in w2])
    a_c = ip(*[w.shape[1] for w in w2])
    a_s = ip(*([0] * n if scaled_rows is None else [int(v) for v in scaled_rows]))
    _timed_call("maest_cast_weights_multi", 0.0, n, ctypes.cast(a_src, ctypes.c_void_p), ctypes.cast(a_dst, ctypes.c_void_p),
                ctypes.cast(a_dt, ctypes.c_void_p), ctypes.cast(a_r, ctypes.c_void_p), ctypes.cast(a_c, ctypes.c_void_p),
                ctypes.cast(a_s, ctypes.c_void_p), float(row_scale), SPLIT3_B if s3 else DT[dtype], _s(srcs[0]))
    return outs


def layernorm_fwd(x: torch.Tensor, gamma, beta, eps: float, out_dtype, save_stats=False):
    """x fp32 [rows, 768] -> y (out_dtype), optionally (mean, rstd)."""
    _chk(x, gamma, beta)
    assert x.dtype == torch.float32 and x.dim() == 2
    rows, cols = x.shape
    s3 = out_dtype == SPLIT3
    y = torch.empty((rows, 3 * cols if s3 else cols), dtype=torch.bfloat16 if s3 else out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    _timed_call("maest_layernorm_fwd", 0.0, _p(x), x.stride(0), _p(gamma), _p(beta), _p(y), y.stride(0), SPLIT3_A if s3 else DT[out_dtype],
                _p(mean), _p(rstd), rows, cols, eps, _s(x))
    return (y, mean, rstd) if save_stats else y


def add_layernorm_fwd(x: torch.Tensor, delta: torch.Tensor, gamma, beta, eps: float, out_dtype, save_stats=False):
    """x fp32 [rows, 768] + delta (any operand dtype) -> (x_new fp32, y (out_dtype)[, mean, rstd])."""
    _chk(x, delta, gamma, beta)
    assert x.dtype == torch.float32 and x.dim() == 2 and delta.shape == x.shape
    rows, cols = x.shape
    x_new = torch.empty_like(x)
    s3 = out_dtype == SPLIT3
    y = torch.empty((rows, 3 * cols if s3 else cols), dtype=torch.bfloat16 if s3 else out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    _timed_call("maest_layernorm_fwd", 0.0, _p(x), _p(delta), DT[delta.dtype], _p(x_new), _p(gamma), _p(beta), _p(y),
                SPLIT3_A if s3 else DT[out_dtype], _p(mean), _p(rstd), rows, cols, eps, _s(x), _entry="maest_add_layernorm_fwd")
    return (x_new, y, mean, rstd) if save_stats else (x_new, y)


def layernorm_bwd(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, lp_dtype=None, want_fp32=True, head_tokens=None):
    """-> (dx fp32 or None, dx in lp_dtype or None); dgamma/dbeta accumulated in place.
    head_tokens = (n_tok, n_head): `dres` is compact, [rows / n_tok * n_head, 768] -- the residual gradient of the
    first n_head tokens of every clip; the other tokens have none."""
    _chk(dy, x, gamma, mean, rstd, dres, dgamma, dbeta)
    rows, cols = x.shape
    dx = torch.empty((rows, cols), dtype=torch.float32, device=x.device) if want_fp32 else None
    dx_lp = torch.empty((rows, cols), dtype=lp_dtype, device=x.device) if lp_dtype is not None else None
    n_tok, n_head = head_tokens if head_tokens is not None else (1, 0)
    if n_head:
        assert dres is not None and dres.shape == (rows // n_tok * n_head, cols) and dres.is_contiguous()
    _timed_call("maest_layernorm_bwd", 0.0, _p(dy), dy.stride(0), DT[dy.dtype], _p(x), x.stride(0), _p(gamma), _p(mean),
                _p(rstd), _p(dres), _p(dx), _p(dx_lp), DT[lp_dtype] if lp_dtype is not None else 0, _p(dgamma), _p(dbeta),
                rows, cols, n_tok, n_head, _s(x), _entry="maest_layernorm_bwd_headres")
    return dx, dx_lp


def _attn_flops(B, N, q_rows, per_pair):
    nq = N if q_rows is None or q_rows >= N else min(N, -(-q_rows // 32) * 32)
    return per_pair * B * HEADS * nq * N * HEAD_DIM


def attn_fwd(qkv: torch.Tensor, B: int, N: int, scale: float, save_lse=False, q_rows=None, x3: bool = False, q_prescaled=False,
             out_split3=False):
    """q_rows: only the first q_rows queries of every clip are wanted (rows beyond the 32-row tile that holds them are
    left unwritten in `out` / `lse`).  q_prescaled (bf16 only): the q columns hold scale * log2(e) * q (MAEST_BF16_QS)."""
    _chk(qkv)
    assert not q_prescaled or (qkv.dtype == torch.bfloat16 and not x3)
    assert qkv.shape == (B * N, 3 * EMBED)
    # out_split3 (x3 mode): the fp32 result leaves as MAEST_SPLIT3_A rows, bf16 [B * N, 3 * 768] -- the A operand of the proj GEMM
    assert not out_split3 or (x3 and qkv.dtype == torch.float32)
    out = (torch.empty((B * N, 3 * EMBED), dtype=torch.bfloat16, device=qkv.device) if out_split3
           else torch.empty((B * N, EMBED), dtype=qkv.dtype, device=qkv.device))
    lse = torch.empty((B, HEADS, N), dtype=torch.float32, device=qkv.device) if save_lse else None
    _timed_call("maest_attn_fwd", _attn_flops(B, N, q_rows, 4.0), _p(qkv), _p(out), _p(lse), B, N,
                F32X3_A3 if out_split3 else (BF16_QS if q_prescaled else _mm_code(qkv.dtype, x3)), scale,
                N if q_rows is None else q_rows, _s(qkv), _entry="maest_attn_fwd_rows")
    return (out, lse) if save_lse else out


def attn_bwd_rows_supported(dtype, N: int) -> bool:
    """Whether maest_attn_bwd_rows serves q_rows < N for this shape (the fused bf16 kernel: include/maest_hip.h)."""
    return dtype == torch.bfloat16 and -(-N // 32) + 2 <= 12 and get_option("attn_bwd") in (0, 3)


def attn_bwd(qkv, out, dout, lse, B: int, N: int, scale: float, q_rows=None, x3: bool = False, delta=None, q_prescaled=False):
    """`delta` (fp32 [B, 12, N] = rowsum(dO * O), from gemm_nt_rowdot) given: `out` is not needed (pass None).
    q_prescaled (bf16 only): the q columns of `qkv` hold scale * log2(e) * q (MAEST_BF16_QS); dQ is still d / d(true q)."""
    _chk(qkv, out, dout, lse, delta)
    assert not q_prescaled or (qkv.dtype == torch.bfloat16 and not x3)
    assert dout.dtype == qkv.dtype and (out is None or out.dtype == qkv.dtype)
    if delta is None:
        assert out is not None
        delta = torch.empty((B, HEADS, N), dtype=torch.float32, device=qkv.device)
    else:
        assert delta.shape == (B, HEADS, N) and delta.dtype == torch.float32 and delta.is_contiguous()
        out = None
    dqkv = torch.empty_like(qkv)
    _timed_call("maest_attn_bwd", _attn_flops(B, N, q_rows, 10.0), _p(qkv), _p(out), _p(dout), _p(lse),
                _p(delta), _p(dqkv), B, N, BF16_QS if q_prescaled else _mm_code(qkv.dtype, x3), scale,
                N if q_rows is None else q_rows, _s(qkv),
                _entry="maest_attn_bwd_rows")
    return dqkv


def gather_head_rows(x: torch.Tensor, clips: int, n_tok: int, n_head: int) -> torch.Tensor:
    """[clips * n_tok, 768] -> the first n_head token rows of every clip, compact [clips * n_head, 768]."""
    _chk(x)
    assert x.shape == (clips * n_tok, EMBED) and x.is_contiguous() and x.dtype in DT
    out = torch.empty((clips * n_head, EMBED), dtype=x.dtype, device=x.device)
    call("maest_gather_head_rows", _p(x), clips, n_tok, n_head, DT[x.dtype], _p(out), _s(x))
    return out


def scatter_head_rows(xc: torch.Tensor, clips: int, n_tok: int, n_head: int, n_pad: int) -> torch.Tensor:
    """compact [clips * n_head, 768] -> a [clips * n_tok, 768] buffer whose rows [0, n_pad) of every clip hold the compact
    rows followed by zeros; the other rows are UNINITIALISED (never read by maest_attn_bwd_rows)."""
    _chk(xc)
    assert xc.shape == (clips * n_head, EMBED) and xc.is_contiguous()
    out = torch.empty((clips * n_tok, EMBED), dtype=xc.dtype, device=xc.device)
    call("maest_scatter_head_rows", _p(xc), clips, n_tok, n_head, n_pad, DT[xc.dtype], _p(out), _s(xc))
    return out


def patch_im2col(x: torch.Tensor, tok_ft: torch.Tensor, dtype, perm=None, lam=None, t_stripes=None, f_stripes=None, stride=(10, 10)):
    """x fp32 or fp16 [B, F, T], tok_ft int32 [P, 2] -> im2col operand [B*P, 256] (SpecMasking stripes, mixup and
    patchout fused; a float16 batch -- what the reference's loader hands out, discogs/dataset.py:58-67 -- is widened in
    the load).  t_stripes / f_stripes: int32 [B, n, 2] = (start, width) per clip, or None.  stride: (frequency, time) step of the
    16 x 16 patches (models/maest.py:214-241)."""
    _chk(x, tok_ft, perm, lam, t_stripes, f_stripes)
    assert x.dtype in (torch.float32, torch.float16) and x.dim() == 3 and tok_ft.dtype == torch.int32
    B, F, T = x.shape
    P = tok_ft.shape[0]
    n_t = n_f = 0
    if t_stripes is not None:
        assert t_stripes.dtype == torch.int32 and t_stripes.dim() == 3 and t_stripes.shape[0] == B and t_stripes.shape[2] == 2
        n_t = int(t_stripes.shape[1])
    if f_stripes is not None:
        assert f_stripes.dtype == torch.int32 and f_stripes.dim() == 3 and f_stripes.shape[0] == B and f_stripes.shape[2] == 2
        n_f = int(f_stripes.shape[1])
    out = torch.empty((B * P, 256), dtype=dtype, device=x.device)
    call("maest_patch_im2col_strided", _p(x), F16 if x.dtype == torch.float16 else F32, B, F, T, int(stride[0]), int(stride[1]), _p(perm), _p(lam),
         _p(tok_ft), P, _p(t_stripes) if n_t else None, n_t, _p(f_stripes) if n_f else None, n_f, _p(out), DT[dtype], _s(x))
    return out


def token_assemble(patches, cls_token, dist_token, new_pos, freq_pos, time_pos, toffset, tok_ft, B):
    _chk(patches, cls_token, dist_token, new_pos, freq_pos, time_pos, tok_ft)
    Tt, Fg, P = time_pos.shape[-1], freq_pos.shape[-1], tok_ft.shape[0]
    x0 = torch.empty((B, 2 + P, EMBED), dtype=torch.float32, device=patches.device)
    call("maest_token_assemble", _p(patches), _p(cls_token), _p(dist_token), _p(new_pos), _p(freq_pos),
         _p(time_pos), Fg, Tt, toffset, _p(tok_ft), B, P, _p(x0), _s(patches))
    return x0


def token_assemble_bwd(dx0, B, Fg, Tt, toffset, tok_ft, dtype, d_cls, d_dist, d_new_pos, d_freq_pos, d_time_pos,
                       want_dpatches=True):
    _chk(dx0, tok_ft, d_cls, d_dist, d_new_pos, d_freq_pos, d_time_pos)
    P = tok_ft.shape[0]
    dp = torch.empty((B * P, EMBED), dtype=dtype, device=dx0.device) if want_dpatches else None
    call("maest_token_assemble_bwd", _p(dx0), B, P, Fg, Tt, toffset, _p(tok_ft), _p(dp), DT[dtype], _p(d_cls),
         _p(d_dist), _p(d_new_pos), _p(d_freq_pos), _p(d_time_pos), _s(dx0))
    return dp


def head_pool_fwd(x: torch.Tensor, gamma, beta, eps: float, save_stats=False):
    """x fp32 [B, N, 768] -> cls, dist, feat (fp32 [B,768]) [, mean, rstd fp32 [B,2]]."""
    _chk(x, gamma, beta)
    B, N, _ = x.shape
    cls = torch.empty((B, EMBED), dtype=torch.float32, device=x.device)
    dist = torch.empty_like(cls)
    feat = torch.empty_like(cls)
    mean = torch.empty((B, 2), dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty((B, 2), dtype=torch.float32, device=x.device) if save_stats else None
    call("maest_head_pool_fwd", _p(x), B, N, _p(gamma), _p(beta), eps, _p(cls), _p(dist), _p(feat), _p(mean),
         _p(rstd), _s(x))
    return (cls, dist, feat, mean, rstd) if save_stats else (cls, dist, feat)


def head_pool_bwd(d_cls, d_dist, d_feat, x, gamma, mean, rstd, dgamma, dbeta):
    _chk(d_cls, d_dist, d_feat, x, gamma, mean, rstd, dgamma, dbeta)
    B, N, _ = x.shape
    dx = torch.empty_like(x)
    call("maest_head_pool_bwd", _p(d_cls), _p(d_dist), _p(d_feat), _p(x), B, N, _p(gamma), _p(mean), _p(rstd),
         _p(dx), _p(dgamma), _p(dbeta), _s(x))
    return dx


def embed_pool(x: torch.Tensor):
    _chk(x)
    B, N, _ = x.shape
    emb = torch.empty((B, 3 * EMBED), dtype=torch.float32, device=x.device)
    call("maest_embed_pool", _p(x), B, N, _p(emb), _s(x))
    return emb


def bce_logits(z, y, weight=1.0, perm=None, lam=None, loss=None, want_grad=True):
    """loss (fp32 scalar tensor, accumulated) and dlogits."""
    _chk(z, y, perm, lam, loss)
    rows, cols = z.shape
    if loss is None:
        loss = torch.zeros((), dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    call("maest_bce_logits", _p(z), _p(y), _p(perm), _p(lam), rows, cols, weight, _p(loss), _p(dz), _s(z))
    return loss, dz


def sigmoid_mean(z: torch.Tensor):
    _chk(z)
    rows, cols = z.shape
    act = torch.empty(cols, dtype=torch.float32, device=z.device)
    call("maest_sigmoid_mean", _p(z), rows, cols, _p(act), _s(z))
    return act


def colsum(src: torch.Tensor, out: torch.Tensor):
    _chk(out)
    rows, cols = src.shape
    _timed_call("maest_colsum", 0.0, _p(src), src.stride(0), rows, cols, DT[src.dtype], _p(out), _s(src))
    return out


def spec_mask_(x: torch.Tensor, t_stripes=None, f_stripes=None):
    """in place; x fp32 [B, F, T]; stripes int32 [B, n, 2] = (start, width)."""
    _chk(x, t_stripes, f_stripes)
    B, F, T = x.shape
    n_t = 0 if t_stripes is None else t_stripes.shape[1]
    n_f = 0 if f_stripes is None else f_stripes.shape[1]
    call("maest_spec_mask", _p(x), B, F, T, _p(t_stripes), n_t, _p(f_stripes), n_f, _s(x))
    return x


def melfile_assemble(frames: torch.Tensor, row_start: torch.Tensor, frames_read: torch.Tensor, T: int,
                     normalize: bool = True, norm_mean: float = 2.06755686098554,
                     norm_std: float = 1.268292820667291) -> torch.Tensor:
    """Raw on-disk mel rows (float16 [rows, n_bands], device) -> network input fp32 [B, n_bands, T]
    (pad + centre-roll + transpose + the datamodule's float16 normalisation; csrc/embed.hip)."""
    _chk(frames, row_start, frames_read)
    assert frames.dtype == torch.float16 and frames.dim() == 2 and frames.is_contiguous()
    assert row_start.dtype == torch.int64 and frames_read.dtype == torch.int32
    B = int(frames_read.numel())
    n_bands = int(frames.shape[1])
    out = torch.empty((B, n_bands, T), dtype=torch.float32, device=frames.device)
    # numpy evaluates (x - mean) / (std * 2) in float16 with both python scalars cast to float16 first
    mean_h = float(np.float16(norm_mean))
    div_h = float(np.float16(norm_std * 2))
    call("maest_melfile_assemble", _p(frames), _p(row_start), _p(frames_read), B, n_bands, T, 1 if normalize else 0,
         mean_h, div_h, _p(out), _s(frames))
    return out


def logmel(wave: torch.Tensor, consts) -> torch.Tensor:
    """wave fp32 [B, S] -> [B, 96, 1 + S // 256].  `consts` = melspectrogram.MelConstants on wave.device."""
    _chk(wave)
    assert wave.dtype == torch.float32 and wave.dim() == 2
    B, S = wave.shape
    T = 1 + S // 256
    out = torch.empty((B, 96, T), dtype=torch.float32, device=wave.device)
    _timed_call("maest_logmel", 0.0, _p(wave), B, S, _p(consts.window), _p(consts.twiddle), _p(consts.fb_start),
         _p(consts.fb_len), _p(consts.fb_w), consts.fb_stride, consts.log_scale, consts.norm_mean,
         consts.norm_2std, _p(out), _s(wave))
    return out


def scale_(x: torch.Tensor, alpha: float):
    _chk(x)
    assert x.dtype == torch.float32
    call("maest_scale_f32", _p(x), x.numel(), alpha, _s(x))
    return x


def scale_dev_(x: torch.Tensor, alpha: torch.Tensor):
    """x *= alpha in place, `alpha` a one-element fp32 DEVICE tensor (no host round trip)."""
    _chk(x, alpha)
    assert x.dtype == torch.float32 and alpha.dtype == torch.float32 and alpha.numel() == 1
    call("maest_scale_dev_f32", _p(x), x.numel(), _p(alpha), _s(x))
    return x


def cast_rows(src: torch.Tensor, dtype, ld_dst: Optional[int] = None) -> torch.Tensor:
    """fp32 [rows, cols] -> dtype [rows, ld_dst], columns cols..ld_dst zero filled."""
    _chk(src)
    assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
    rows, cols = src.shape
    ld_dst = ld_dst or cols
    out = torch.empty((rows, ld_dst), dtype=dtype, device=src.device)
    call("maest_cast_rows", _p(src), src.stride(0), _p(out), ld_dst, rows, cols, DT[dtype], _s(src))
    return out


def affine_(x: torch.Tensor, add: float, div: float):
    """x = (x + add) / div in place (fp32)."""
    _chk(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    call("maest_affine_f32", _p(x), x.numel(), add, div, _s(x))
    return x


def augment_mel(wave, window, twiddle, fb_start, fb_len, fb_w, fb_stride, n_mels, pre0, pre1, log_eps, norm_add, norm_div):
    """wave fp32 [B, S] (32 kHz) -> fp32 [B, n_mels, 1 + (S - 1) // 320]  (csrc/mel2.hip)."""
    _chk(wave, window, twiddle, fb_start, fb_len, fb_w)
    assert wave.dtype == torch.float32 and wave.dim() == 2 and wave.is_contiguous()
    B, S = wave.shape
    T = 1 + (S - 1) // 320
    out = torch.empty((B, n_mels, T), dtype=torch.float32, device=wave.device)
    call("maest_augment_mel", _p(wave), B, S, _p(window), _p(twiddle), _p(fb_start), _p(fb_len), _p(fb_w), fb_stride,
         n_mels, pre0, pre1, log_eps, norm_add, norm_div, _p(out), _s(wave))
    return out


def swa_update_multi(avgs, curs, inv_count: float):
    """avg += (cur - avg) * inv_count for every (avg, cur) pair of fp32 tensors, one launch."""
    _chk(*avgs, *curs)
    n = len(avgs)
    assert n == len(curs) and n > 0
    for a, c in zip(avgs, curs):
        assert a.dtype == c.dtype == torch.float32 and a.is_contiguous() and c.is_contiguous() and a.numel() == c.numel()
    vp = ctypes.c_void_p * n
    a_avg = vp(*[a.data_ptr() for a in avgs])
    a_cur = vp(*[c.data_ptr() for c in curs])
    a_n = (ctypes.c_int64 * n)(*[a.numel() for a in avgs])
    call("maest_swa_update_multi", n, ctypes.cast(a_avg, ctypes.c_void_p), ctypes.cast(a_cur, ctypes.c_void_p),
         ctypes.cast(a_n, ctypes.c_void_p), float(inv_count), _s(avgs[0]))


# ------------------------------------------------------------------------------------ process-wide switches
def set_option(name: str, value: Optional[int]):
    """maest_set_option: `value` None restores the default (environment, read once at first use)."""
    opt = _lib.OPTIONS[name]
    call("maest_set_option", opt, 0 if value is None else int(value), 1 if value is None else 0)
    _option_cache.pop((id(_lib.load()), name), None)


# library switches change only through set_option (their environment defaults are read once by the library): the hot path
# (one query per wgrad GEMM and per block of a backward pass) reads them from here instead of crossing the C ABI each time
_option_cache = {}


def get_option(name: str) -> int:
    key = (id(_lib.load()), name)
    v = _option_cache.get(key)
    if v is None:
        c = ctypes.c_int(0)
        call("maest_get_option", _lib.OPTIONS[name], ctypes.byref(c))
        v = _option_cache[key] = c.value
    return v


class thread_options:
    """``with ops.thread_options(gemm_wgs=256): ...`` -- override switches for launches made by THIS thread inside the block
    (maest_set_option_thread); other threads, and this thread afterwards, see the process-wide values.  Not re-entrant per switch."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            call("maest_set_option_thread", _lib.OPTIONS[k], int(v), 0)
        return self

    def __exit__(self, *a):
        for k in self.kw:
            call("maest_set_option_thread", _lib.OPTIONS[k], 0, 1)


class options:
    """``with ops.options(gemm_min_m=512): ...`` -- set switches for a block, restore the previous values after."""

    def __init__(self, **kw):
        self.kw = kw
        self.prev = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.prev.items():
            set_option(k, v)
