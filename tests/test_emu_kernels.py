"""CPU (`-m "not gpu"`): run the SAME kernel sources under the SIMT lockstep emulator (tests/emu) at
tiny shapes and compare with the oracle.  Checks index arithmetic, LDS layouts, MFMA fragment
bookkeeping, tail masking and barrier placement -- not timing."""
import pytest
import torch

from tests import kernel_cases as KC

DT = [torch.float32, torch.bfloat16]
# The fp32-MFMA instantiations of the 256-row-tile GEMM kernels take 30-50 s each under the emulator (K = 2 per MFMA): the CPU suite runs
# their bf16 instantiations (same template, same ring / stagger / epilogue code); fp32 runs on the GPU (tests/test_kernels_gpu.py:
# test_gemm_big_tile_kernels_at_small_m, test_gemm_128_row_tiles_forced, test_gemm_tn).
DT_BIG = [torch.bfloat16]


@pytest.mark.parametrize("dtype", DT)
def test_emu_gemm(emu, dtype):
    K = 64 if dtype == torch.float32 else 128
    KC.case_gemm(emu, dtype, 150, 200, K)


@pytest.mark.parametrize("dtype", DT)
def test_emu_gemm_ragged_n_scalar_epilogue(emu, dtype):
    K = 32 if dtype == torch.float32 else 64
    KC.case_gemm(emu, dtype, 70, 51, K)


@pytest.mark.parametrize("dtype", DT_BIG)
def test_emu_gemm_256_tile_full_line_stages(emu, dtype, gemm_options):
    """gemm_nt256w_kernel (M >= 512, N % 256 == 0; fp32 / bf16x3 operands, and bf16 under gemm_variant = 3): 128-byte K stages through a 5-buffer unit
    ring (6 stages: the ring wraps)."""
    gemm_options(gemm_min_m=512, gemm_variant=3)   # (the bf16 default is gemm_nt256o_kernel: test_emu_gemm_one_wave_per_simd_kernel)
    KC.case_gemm(emu, dtype, 512, 256, 192 if dtype == torch.float32 else 384, identity=False)   # 6 stages > 5 buffers


def test_emu_gemm_256_tile_full_line_stages_fp32_instantiation(emu, gemm_options):
    """One small case of the fp32 instantiation of gemm_nt256w_kernel / gemm_tn256_kernel (the production path of precision "fp32"
    and, with three bf16 MFMAs per product, of the default "bf16x3" inference) so that a CPU-only run exercises that template too:
    K = 64 fp32 = two 128-byte stages; one 256 x 256 wgrad tile over 64 tokens."""
    gemm_options(gemm_min_m=512)
    KC.case_gemm(emu, torch.float32, 512, 256, 64, identity=False)
    gemm_options(gemm_variant=4)
    KC.case_gemm_tn(emu, torch.float32, 64, 256, 256)


def test_emu_gemm_one_wave_per_simd_kernel(emu, gemm_options):
    """gemm_nt256o_kernel's host twin (the C++ form of every owned-register primitive of gemm_nt_ow.hip): 1, 2, 3 and 7 K stages
    (the prologue's three forms, the stage kinds FIRST / full / last-but-one / last, two ring phases beyond the period) and a
    ragged second tile row, bit for bit against the 8-wave kernel in every epilogue form; then several tiles per workgroup."""
    gemm_options(gemm_min_m=512, gemm_tail=0)
    KC.case_gemm_one_wave_per_simd(emu, 512, 256, 64, only=("none -> bf16",), pair=False)
    KC.case_gemm_one_wave_per_simd(emu, 512, 256, 128, only=("residual -> fp32",), pair=False)
    KC.case_gemm_one_wave_per_simd(emu, 512, 256, 192, only=("mul -> bf16", "gelu -> fp32"), pair=False)
    KC.case_gemm_one_wave_per_simd(emu, 520, 256, 448, only=("none -> fp32",))
    # the persistent tile loop: ONE workgroup walks the three tiles (next tile's A_0 / B_0 requested from inside the epilogue -- at its
    # start, and behind the last pass's operand fetch in the RESIDUAL form --, A_1 / B_1 behind it, stage 0 without its own A_0 / B_0)
    gemm_options(gemm_wgs=1)
    KC.case_gemm_one_wave_per_simd(emu, 520, 256, 128, only=("none -> bf16", "residual -> fp32"), pair=False)
    # tile order in column panels (tile_of): 2 x 3 tiles walked as panels of 2 + 1
    gemm_options(gemm_wgs=0, gemm_panel=2)
    KC.case_gemm_one_wave_per_simd(emu, 512, 768, 64, only=("none -> bf16",), pair=False)


@pytest.mark.parametrize("dtype", DT_BIG)
def test_emu_gemm_128_row_tiles_of_the_last_partial_round(emu, dtype, gemm_options):
    """gemm_nt256w_kernel<MTW = 2>: 128 x 256 tiles (A units of 128 rows in the same ring, vmcnt(2) waits, one-pass
    epilogue with the aux operand prefetched), forced onto every tile; 576 rows = 4.5 tiles (ragged last tile)."""
    gemm_options(gemm_min_m=512, gemm_tail=2)
    KC.case_gemm(emu, dtype, 576, 256, 192 if dtype == torch.float32 else 384, identity=False)


@pytest.mark.parametrize("dtype", DT)
def test_emu_gemm_rowdot(emu, dtype, gemm_options):
    """the row-dot side output: stand-alone reduction behind the 128x128 kernel, the two-pass epilogue of the full-line
    kernel, and the one-pass epilogue of its 128-row-tile variant"""
    K = 64 if dtype == torch.float32 else 128
    KC.case_gemm_rowdot(emu, dtype, 150, 128, K, 75)
    gemm_options(gemm_min_m=512)
    KC.case_gemm_rowdot(emu, dtype, 512, 256, K, 64)
    gemm_options(gemm_min_m=512, gemm_tail=2)
    KC.case_gemm_rowdot(emu, dtype, 576, 256, K, 96)


@pytest.mark.parametrize("dtype", DT_BIG)
def test_emu_gemm_tn_256_tile_lds_dma_kernel(emu, dtype, gemm_options):
    """M, N multiples of 256 and K a multiple of the slice route to gemm256.hip:gemm_tn256_kernel."""
    gemm_options(gemm_variant=4)   # take the 256-tile kernel although there are only 2 tiles
    KC.case_gemm_tn(emu, dtype, 288 if dtype == torch.bfloat16 else 144, 256, 512)


def test_emu_gemm_tn_one_wave_per_simd_kernel(emu, gemm_options):
    """gemm_tn256o_kernel's host twin (gemm_tn_ow.hip; the bf16 case of the test above runs it with 9 and 3 slices per workgroup):
    5 and 4 slices (split_k = 2 of 9: the prologue's A-half-of-slice-4 form and the short one, slice kinds 1 and 0 first) and 12
    (two trips round the five-buffer ring).  The workspace form inside case_gemm_tn keeps running the 8-wave kernel."""
    gemm_options(gemm_variant=4)
    KC.case_gemm_tn(emu, torch.bfloat16, 288, 256, 512, splits=(2,))
    KC.case_gemm_tn(emu, torch.bfloat16, 384, 256, 256, splits=(1,))


@pytest.mark.parametrize("dtype", DT)
def test_emu_gemm_tn(emu, dtype):
    KC.case_gemm_tn(emu, dtype, 150, 136, 200)
    KC.case_gemm_tn(emu, dtype, 40, 24, 72, lda_pad=8)


@pytest.mark.parametrize("dtype", DT)
def test_emu_transpose(emu, dtype):
    KC.case_transpose(emu, dtype, 70, 130)


@pytest.mark.parametrize("dtype", DT)
def test_emu_layernorm(emu, dtype):
    KC.case_layernorm(emu, dtype, 11)


@pytest.mark.parametrize("dtype", DT)
def test_emu_attention(emu, dtype):
    # bf16: two key tiles; fp32 (4x the emulated MFMAs): one tile here, its multi-tile path is the spike case below
    KC.case_attention(emu, dtype, 1, 75 if dtype == torch.bfloat16 else 40)


def test_emu_attention_prescaled_q(emu):
    """MAEST_BF16_QS: q columns pre-multiplied by scale * log2(e); every bf16 form (the persistent forward reads its Q fragments
    straight from the rows) against the oracle on the true q; one ragged key tile (the scale enters per score and in the Q take only:
    the multi-tile walks are the raw-q cases above and the GPU tests)."""
    KC.case_attention(emu, torch.bfloat16, 1, 40, qs=True)


def test_emu_attention_multi_tile_spike(emu):
    KC.case_attention(emu, torch.float32, 1, 100, spike=True)


def test_emu_attention_persistent_forward(emu):
    """attn_fwd_pw_kernel's host twin (the C++ form of every owned-register primitive): 3 key tiles with both fragment-set parities, a
    ragged last tile, the rescale path (spike) -- one item; the multi-item walk is a GPU test."""
    from maest_amd import ops
    import tests.kernel_cases as K
    qkv = K.rnd((130, 2304), 20, 1.0)
    qkv[70, 768:768 + 64] = qkv[3, 0:64] * 6.0
    qkv = qkv.to(torch.bfloat16)
    with ops.options(attn_fwd=3):
        out, lse = ops.attn_fwd(qkv, 1, 130, 0.125, save_lse=True)
    ref, ref_lse = K._attn_ref(qkv.float(), 1, 130, 0.125)
    K.close(out, ref, 2e-2, 2e-2, "persistent attention forward (emulated)")
    K.close(lse, ref_lse, 1e-4, 2e-2, "persistent attention forward lse (emulated)")


def test_emu_split_bf16_products(emu):
    """precision="bf16x3": fp32 tensors, three bf16 MFMAs per product on hi/lo operand splits (GEMM + attention fwd)."""
    KC.case_split_precision(emu, M=512, N=256, K=192, B=1, Ntok=40)


@pytest.mark.parametrize("dtype", DT)
def test_emu_patch_embed(emu, dtype):
    KC.case_patch_embed(emu, dtype, 2, 66, patchout=2, mix=True)


def test_emu_patch_embed_eval(emu):
    KC.case_patch_embed(emu, torch.float32, 1, 56)
    KC.case_patch_embed(emu, torch.float32, 2, 70, patchout=1, mix=True, masked=True, stride=(16, 13), seed=35)      # another patch stride


@pytest.mark.parametrize("mix", [False, True])
def test_emu_patch_embed_spec_masking_fused(emu, mix):
    KC.case_patch_embed(emu, torch.float32, 3, 66, patchout=1, mix=mix, masked=True, seed=33)


def test_emu_head(emu):
    KC.case_head(emu, 3, 7)


def test_emu_loss(emu):
    KC.case_loss(emu, 6, 50)


def test_emu_spec_mask(emu):
    KC.case_spec_mask(emu, 2, 40)


def test_emu_mel(emu):
    KC.case_mel(emu, 1, 2560)


def test_emu_mel_interior_blocks(emu):
    """193 frames = four 64-frame blocks: the two middle ones take the aligned 8-byte fetch form (the packed-math twins of the FFT stages run
    under both forms), the second clip starts off the 8-byte grid (odd length) and takes the element-wise form throughout."""
    KC.case_mel(emu, 2, 49153, seed=85)


def test_emu_melfile(emu, tmp_path):
    KC.case_melfile(emu, tmp_path)


def test_emu_augment_mel(emu):
    KC.case_augment_mel(emu, 1, 4000)


def test_emu_swa(emu):
    KC.case_swa(emu)


@pytest.mark.parametrize("dtype,B,N", [(torch.bfloat16, 2, 75), (torch.bfloat16, 1, 20), (torch.float32, 1, 40)])
def test_attention_restricted_to_the_head_tokens(emu, dtype, B, N):
    KC.case_attention_head_rows(emu, dtype, B, N)
