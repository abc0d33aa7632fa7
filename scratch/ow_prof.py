"""gemm_nt256o_kernel time budget: needs scratch/pw_abl/libmaest_<name>.so built with -DOW_PROF (scratch/ow_ablate.sh "prof:-DOW_PROF").
Prints, for the four waves of workgroup 5, the shader-clock cycles spent per stage in each part of the loop (gemm_nt_ow.hip: OW_TICK)."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import ops, _lib
names = sys.argv[1:] or ["prof"]
M, N, K = 65536, 4096, 4096
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for name in names:
    lib = ctypes.CDLL("scratch/pw_abl/libmaest_%s.so" % name)
    _lib._lib = _lib._bind(lib)
    buf = torch.zeros(192, dtype=torch.int64, device="cuda")
    with ops.options(gemm_tail=0):
        for _ in range(2): ops.gemm_nt(a, w, None, out=o)
        torch.cuda.synchronize()
        lib.maest_debug_ow_prof.argtypes = [ctypes.c_void_p]
        assert lib.maest_debug_ow_prof(buf.data_ptr()) == 0
        ops.gemm_nt(a, w, None, out=o)
        torch.cuda.synchronize()
        lib.maest_debug_ow_prof(None)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with ops.options(gemm_tail=0):
        e0.record(); ops.gemm_nt(a, w, None, out=o); e1.record(); torch.cuda.synchronize()
    allb = buf.cpu().reshape(2, 4, 24).tolist()
    span = allb[1][0][23] - allb[0][0][22]
    print(f"   first block start -> last-round block end: {span} ticks; kernel wall {e0.elapsed_time(e1) * 1e3:.0f} us -> {span / (e0.elapsed_time(e1) * 1e3):.0f} ticks/us")
    for blk, t in enumerate(allb):
        ns = K // 64; print("  block", "5" if blk == 0 else "grid-3")
        lab = ["step3", "wait0", "step0", "wait1", "step1", "wait2", "step2", "wait3", "vmcnt", "barrier", "prologue", "drain", "epilogue"]
        print(f"== {name}: cycles per stage (k-step = 16 MFMAs = 512 matrix-pipe cycles), {ns} stages")
        for wv in range(4):
            per = [t[wv][i] / ns for i in range(10)]
            print(f"  wave {wv}: " + "  ".join(f"{lab[i]} {per[i]:6.0f}" for i in range(10)) + f" | sum {sum(per):6.0f} | prologue {t[wv][10]} drain {t[wv][11]} epilogue: stage {t[wv][13]} sync {t[wv][14]} drain {t[wv][15]} ack {t[wv][12]}")
