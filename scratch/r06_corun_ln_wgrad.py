"""Physics check for a scheduled backward: a TN wgrad GEMM launched W workgroups wide (each owns a whole CU) on one stream, a
LayerNorm backward (HBM-bound, no LDS, few registers) on another, launched right after it -- the pair's wall time against the two alone."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops, maest as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
rows, E, H = 256 * 290, 768, 3072
bf = torch.bfloat16
dy = torch.randn(rows, E, device=dev).to(bf); g = torch.randn(rows, H, device=dev).to(bf)
x = torch.randn(rows, E, device=dev); gamma = torch.randn(E, device=dev)
mean = x.mean(1); rstd = 1.0 / x.std(1)
dres = torch.randn(rows, E, device=dev)
dln = torch.randn(rows, E, device=dev).to(bf)
gw = torch.zeros(E, device=dev); gb = torch.zeros(E, device=dev)
dw = torch.zeros(E, H, device=dev); db = torch.zeros(E, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def run_wgrad(w):
    M._wgrad(dy, g, E, H, dw, db, False, wgs=w)

def run_ln():
    ops.layernorm_bwd(dln, x, gamma, mean, rstd, dres, gw, gb, lp_dtype=bf)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]

def pair(w, ln_first=False, n_ln=1):
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        if ln_first:
            with torch.cuda.stream(s2):
                for _ in range(n_ln): run_ln()
        with torch.cuda.stream(s1): run_wgrad(w)
        if not ln_first:
            with torch.cuda.stream(s2):
                for _ in range(n_ln): run_ln()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f

print("LayerNorm backward alone: %.0f us" % timed(run_ln))
for w in (0, 224, 192, 160, 128, 96):
    ta = timed(lambda: run_wgrad(w))
    tp = timed(pair(w)); tp2 = timed(pair(w, n_ln=2))
    print("wgrad fc2 at %3d workgroups: alone %.0f us; beside one LayerNorm backward %.0f us; beside two %.0f us" % (w, ta, tp, tp2))
print("LN first, then wgrad 192: %.0f us" % timed(pair(192, ln_first=True)))
