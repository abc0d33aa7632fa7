"""Run the attention kernel cases (tests/kernel_cases.py: every bf16 forward / backward form against the fp32 oracle) on a variant library
under scratch/pw_abl (scratch/pw_ablate.sh): usage  pw_variant_check.py <name> [more names]"""
import sys, ctypes, torch
sys.path.insert(0, ".")
from maest_amd import _lib
from tests import kernel_cases as KC
for name in sys.argv[1:]:
    _lib._lib = _lib._bind(ctypes.CDLL("scratch/pw_abl/libmaest_%s.so" % name))
    for (B, N, spike, qs) in [(2, 560, False, True), (2, 560, True, True), (1, 875, True, True), (13, 875, False, True), (1, 1685, False, True),
                              (45, 560, False, True), (2, 321, False, True), (1, 130, True, True), (2, 560, False, False), (1, 64, False, True)]:
        KC.case_attention("cuda", torch.bfloat16, B, N, spike=spike, qs=qs, bf16_tol=8e-2 if spike else 3e-2, fwd_tol=4e-2 if (spike and not qs) else 2e-2)
        print(f"{name}: B={B} N={N} spike={int(spike)} qs={int(qs)} ok", flush=True)
