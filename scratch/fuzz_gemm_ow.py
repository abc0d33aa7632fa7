"""fuzz of the one-wave-per-SIMD GEMM kernels against the 8-wave kernels they replace: random ragged M, N, K, epilogue forms, workgroup
caps and column-panel widths (NT: within one bf16 ulp / 4e-7 sqrt(K) since the 16x16x32 MFMA of round 6 -- tests/kernel_cases.py: _same_products);
random K / split-K (TN: against the fp32 matmul and the 8-wave kernel)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from maest_amd import ops
from tests.kernel_cases import _same_products
dev = "cuda"; dt = torch.bfloat16
rng = np.random.Generator(np.random.PCG64(4))
torch.manual_seed(1)
forms = ["none_bf16", "none_f32", "gelu_bf16", "gelu_f32", "pair", "resid", "mul_bf16", "mul_f32"]
bad = 0
for it in range(60):
    M = int(rng.integers(512, 9000)); N = int(rng.choice([256, 512, 768, 2304, 3072])); K = 64 * int(rng.integers(1, 49))
    form = forms[int(rng.integers(0, len(forms)))]; wgs = int(rng.choice([0, 0, 1, 8, 24, 256])); tail = int(rng.integers(0, 2)); panel = int(rng.choice([0, 0, -1, 1, 2, 5]))
    a = (torch.randn(M, K, device=dev)).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
    bias = torch.randn(N, device=dev) if rng.integers(0, 2) else None
    kw = {}
    odt = dt if form.endswith("bf16") or form == "pair" else torch.float32
    if form.startswith("gelu") or form == "pair": kw["epi"] = ops.EPI_GELU
    if form == "resid": kw.update(epi=ops.EPI_RESIDUAL, aux_in=torch.randn(M, N, device=dev))
    if form.startswith("mul"): kw.update(epi=ops.EPI_MUL, aux_in=torch.randn(M, N, device=dev).to(odt))
    outs = []
    for variant in (0, 3):
        k2 = dict(kw)
        aux = torch.empty(M, N, device=dev, dtype=dt) if form == "pair" else None
        if aux is not None: k2["aux_out"] = aux
        with ops.options(gemm_variant=variant, gemm_min_m=512, gemm_wgs=wgs, gemm_tail=tail, gemm_panel=panel):
            o = ops.gemm_nt(a, w, bias, out_dtype=odt, **k2)
        outs.append((o, aux))
    try:
        _same_products(outs[0][0], outs[1][0], False, "C", K)
        if outs[0][1] is not None: _same_products(outs[0][1], outs[1][1], False, "aux", K)
        ok = True
    except AssertionError as e:
        ok = False; print("   ", e)
    fin = bool(torch.isfinite(outs[0][0].float()).all())
    if not (ok and fin):
        bad += 1
        print(f"NT MISMATCH M={M} N={N} K={K} {form} wgs={wgs} tail={tail} panel={panel} bias={bias is not None} finite={fin}", flush=True)
print("NT fuzz: 60 cases,", bad, "bad", flush=True)
badt = 0
for it in range(24):
    K = 32 * int(rng.integers(8, 600)); M = 256 * int(rng.integers(1, 5)); N = 256 * int(rng.integers(1, 5)); sk = int(rng.choice([0, 1, 2, 3, 7, 13]))
    a = torch.randn(K, M, device=dev).to(dt); b = torch.randn(K, N, device=dev).to(dt)
    ref = a.float().t() @ b.float(); refc = a.float().sum(0)
    res = []
    for variant in (4, 3):
        out = torch.zeros(M, N, device=dev); cs = torch.zeros(M, device=dev)
        with ops.options(gemm_variant=variant):
            ops.gemm_tn(a, b, out, colsum=cs, split_k=sk)
        res.append((out, cs))
    sc = ref.abs().max().item(); e0 = (res[0][0] - ref).abs().max().item() / sc; ec = (res[0][1] - refc).abs().max().item() / max(refc.abs().max().item(), 1e-6)
    if not (e0 < 2e-5 and ec < 2e-5):
        badt += 1
        print(f"TN MISMATCH K={K} M={M} N={N} split={sk}: err {e0:.2e} colsum {ec:.2e}", flush=True)
print("TN fuzz: 24 cases,", badt, "bad", flush=True)
sys.exit(1 if bad or badt else 0)
