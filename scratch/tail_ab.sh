#!/bin/bash
# 128-row tail tiles (MAEST_GEMM_TAIL = 1, the 8-wave kernel) against a partial last round of the one-wave-per-SIMD kernel (0), in the step
for w in 1 0 1 0; do MAEST_GEMM_TAIL=$w python bench.py --steps 10 --warmup 3 --no-side-cases --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('train tail $w', d['value'], d['ms_per_step'], d['roofline']['frac'], k['maest_gemm_nt'], k['maest_attn_bwd'])"; done
for w in 1 0 1 0; do MAEST_GEMM_TAIL=$w python bench.py --mode infer --steps 20 --warmup 3 --no-side-cases --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('infer tail $w', d['value'], d['ms_per_step'], d['roofline']['frac'], k['maest_gemm_nt'])"; done
