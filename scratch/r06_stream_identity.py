"""Which of torch's pool streams run BESIDE the caller's (default) stream: a narrow TN weight-gradient GEMM (108 workgroups, a whole CU each) on
pool stream k, a full-width NT GEMM on the default stream launched right after it -- together vs alone."""
import sys, torch
sys.path.insert(0, ".")
from maest_amd import ops, maest as M

dev = torch.device("cuda:0")
torch.manual_seed(0)
rows, E, H = 256 * 290, 768, 3072
bf = torch.bfloat16
dy = torch.randn(rows, E, device=dev).to(bf); g = torch.randn(rows, H, device=dev).to(bf)
w = torch.randn(E, H, device=dev).to(bf)
dw = torch.zeros(E, H, device=dev); db = torch.zeros(E, device=dev)

def wgrad(): M._wgrad(dy, g, E, H, dw, db, False, wgs=128)
def dgrad(): ops.gemm_nt(g, w, None, out_dtype=bf)

def timed(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]

with ops.thread_options(gemm_wgs=256, gemm_tail=0):
    print("alone: wgrad at 108 workgroups %.0f us, NT GEMM %.0f us" % (timed(wgrad), timed(dgrad)))
    pool = [torch.cuda.Stream() for _ in range(12)]
    def pair_on(s):
        def pair():
            cur = torch.cuda.current_stream()
            s.wait_stream(cur)
            with torch.cuda.stream(s): wgrad()
            dgrad()
            cur.wait_stream(s)
        return pair
    for order in (range(12), reversed(range(12)), range(12)):
        print("together, pool streams " + " ".join("%d:%.0f" % (k, timed(pair_on(pool[k]))) for k in order), flush=True)
    hp = torch.cuda.Stream(priority=-1)
    print("high-priority stream (stream_id %d): together %.0f us" % (hp.stream_id, timed(pair_on(hp))))
    print("ids: " + " ".join("%d:%d" % (k, pool[k].stream_id) for k in range(12)))
