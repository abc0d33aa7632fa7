"""GPU (`-m gpu`): every C-ABI kernel vs the oracle on a real MI355X, at the shapes the model uses."""
import numpy as np
import pytest
import torch

from tests import kernel_cases as KC

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]
DEV = "cuda"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1120, 2304, 768), (562, 768, 3072), (300, 400, 768), (2 * 9 * 32, 768, 256),
                                   (1024, 2304, 768), (2560, 768, 3072), (9216, 768, 256), (74240, 768, 768)])
def test_gemm(dtype, shape):
    KC.case_gemm(DEV, dtype, *shape)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1120, 2304, 768), (2560, 768, 3072), (1024, 768, 256)])
def test_gemm_big_tile_kernels_at_small_m(dtype, shape, gemm_options):
    """the 256-row-tile kernels (normally taken from 8192 rows up) forced onto small and ragged M"""
    gemm_options(gemm_min_m=512)
    KC.case_gemm(DEV, dtype, *shape)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1120, 2304, 768), (2560, 768, 3072)])
def test_gemm_128_row_tiles_forced(dtype, shape, gemm_options):
    """gemm_nt256w_kernel<MTW = 2> (the kernel of the last partial round) on every tile, every epilogue"""
    gemm_options(gemm_min_m=512, gemm_tail=2)
    KC.case_gemm(DEV, dtype, *shape)


@pytest.mark.parametrize("shape", [(512, 256, 64), (512, 256, 128), (777, 512, 192), (1120, 2304, 768), (2560 + 77, 768, 3072),
                                   (74240, 768, 768), (66000, 2304, 768)])
def test_gemm_one_wave_per_simd_kernel(shape, gemm_options):
    """gemm_nt256o_kernel (bf16, the default 256 x 256 kernel) bit for bit against the 8-wave kernel in every epilogue form: 1 / 2 / 3 /
    12 / 48 K stages, ragged last tile rows, the production shapes (incl. the split into full rounds + 128-row tail tiles)"""
    gemm_options(gemm_min_m=512)
    KC.case_gemm_one_wave_per_simd(DEV, *shape)


@pytest.mark.parametrize("wgs", [1, 8, 24])
def test_gemm_one_wave_per_simd_kernel_walks_tiles(wgs, gemm_options):
    """the persistent tile loop of gemm_nt256o_kernel with 33 tiles on 1 / 8 / 24 workgroups (33, 5 and 2 tiles per workgroup, ragged
    last tile row): every epilogue form bit for bit against the 8-wave kernel"""
    gemm_options(gemm_min_m=512, gemm_tail=0, gemm_wgs=wgs)
    KC.case_gemm_one_wave_per_simd(DEV, 2560 + 77, 768, 768)


@pytest.mark.parametrize("panel", [1, 2, 5, 0])
@pytest.mark.parametrize("wgs", [0, 24])
def test_gemm_one_wave_per_simd_kernel_column_panels(panel, wgs, gemm_options):
    """gemm_nt256o_kernel's tile order (gemm_nt_ow.hip: tile_of): 11 x 9 tiles walked in column panels of 1 / 2 (last panel narrower) /
    5 + 4 tiles and row-major, one workgroup per tile and 24 persistent workgroups: a permutation of the tiles, so every epilogue form
    stays bit-equal to the 8-wave kernel (which has no panels)"""
    gemm_options(gemm_min_m=512, gemm_tail=0, gemm_panel=panel, gemm_wgs=wgs)
    KC.case_gemm_one_wave_per_simd(DEV, 2560 + 77, 2304, 768, only=("none -> bf16", "mul -> bf16", "residual -> fp32"))


def test_gemm_column_panels_chosen_for_the_wide_shapes(gemm_options):
    """the automatic choice (gemm_panel = -1) at the shapes it changes -- fc1 (N = 3072, K = 768: panels of 6) and qkv (N = 2304: 5 + 4) at
    8192 rows -- against row-major order"""
    for N in (3072, 2304):
        a = KC.rnd((8192, 768), 5).to(torch.bfloat16).to(DEV)
        w = (KC.rnd((N, 768), 6) * 0.1).to(torch.bfloat16).to(DEV)
        from maest_amd import ops
        auto = ops.gemm_nt(a, w, None, out_dtype=torch.bfloat16)
        gemm_options(gemm_panel=0)
        assert torch.equal(auto, ops.gemm_nt(a, w, None, out_dtype=torch.bfloat16))
        gemm_options(gemm_panel=-1)


def test_gemm_eight_wave_kernel_still_serves_bf16(gemm_options):
    """gemm_variant = 3: gemm_nt256w_kernel<bf16> (the A/B reference of the kernel above) against the oracle on its own"""
    gemm_options(gemm_min_m=512, gemm_variant=3)
    KC.case_gemm(DEV, torch.bfloat16, 1120, 2304, 768)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_last_partial_round_split(dtype):
    """66000 x 768: 774 tiles of 256 rows = 3 rounds + 6 tiles -> rows 0..65535 in 256-row tiles, the remaining 464
    rows (3.6 tiles of 128, ragged) in a second launch through offset pointers -- every epilogue and its aux operand"""
    KC.case_gemm(DEV, dtype, 66000, 768, 768, identity=False)


@pytest.mark.parametrize("dtype", DT)
def test_gemm_rowdot(dtype, gemm_options):
    """delta = rowsum(dO * O) out of the proj dgrad GEMM: the bench shape (74240 x 768 x 768, 290 tokens per clip: rows
    0..65535 in 256-row tiles + the last partial round in 128-row tiles, whose rows start in the middle of a clip), a
    small shape through the stand-alone reduction, and the 256-row kernels forced onto a small M"""
    KC.case_gemm_rowdot(DEV, dtype, 74240, 768, 768, 290)
    KC.case_gemm_rowdot(DEV, dtype, 1120, 768, 768, 280)
    gemm_options(gemm_min_m=512)
    KC.case_gemm_rowdot(DEV, dtype, 1120, 768, 768, 560)


def test_split_bf16_products_128_row_tiles(gemm_options):
    gemm_options(gemm_tail=2)
    KC.case_split_precision(DEV)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(1121, 768, 3072), (2300, 2304, 768), (64, 400, 768), (7, 519 + 57, 768), (5184, 768, 256), (4640, 768, 3072), (2320, 2304, 768), (2304, 768, 768)])
def test_gemm_tn(dtype, shape):
    K, M, N = shape
    if M == 519 + 57:
        KC.case_gemm_tn(DEV, dtype, K, 519, N, lda_pad=57)
    else:
        KC.case_gemm_tn(DEV, dtype, K, M, N)


@pytest.mark.parametrize("shape", [(74240, 768, 768), (9280, 2304, 768), (8352, 768, 3072), (160 * 32, 3072, 768)])
def test_gemm_tn_one_wave_per_simd_kernel(shape, gemm_options):
    """gemm_tn256o_kernel (bf16, the default 256 x 256 wgrad kernel) at the production shapes, automatic and forced splits, against the
    oracle's fp32 matmul -- and the 8-wave kernel it replaces on the same operands (gemm_variant = 3)"""
    K, M, N = shape
    KC.case_gemm_tn(DEV, torch.bfloat16, K, M, N, splits=(0, 1, 5, 29))
    gemm_options(gemm_variant=3)
    KC.case_gemm_tn(DEV, torch.bfloat16, K, M, N, splits=(0,))


@pytest.mark.parametrize("dtype", DT)
def test_gemm_ragged_n(dtype):
    KC.case_gemm(DEV, dtype, 300, 519, 768)


@pytest.mark.parametrize("dtype", DT)
def test_transpose(dtype):
    KC.case_transpose(DEV, dtype, 1121, 768)
    KC.case_transpose(DEV, dtype, 562, 3072)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm(dtype):
    KC.case_layernorm(DEV, dtype, 1123)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("BN", [(2, 560), (3, 281), (1, 875), (1, 64), (1, 129)])
def test_attention(dtype, BN):
    KC.case_attention(DEV, dtype, *BN)


@pytest.mark.parametrize("BN", [(2, 560), (3, 281), (1, 875), (1, 64), (13, 875), (24, 290), (2, 321), (1, 1685)])
def test_attention_prescaled_q(BN):
    """MAEST_BF16_QS (what the bf16 model runs): q columns pre-multiplied by scale * log2(e) -- every forward / backward form against the
    fp32 oracle on q = q' / (scale * log2 e); the persistent forward takes its Q fragments straight from the rows."""
    KC.case_attention(DEV, torch.bfloat16, *BN, qs=True)


@pytest.mark.parametrize("BN", [(1, 1685), (13, 875), (45, 560), (2, 321), (30, 551)])
def test_attention_persistent_forward_walks_items(BN):
    """attn_fwd_pw_kernel (bf16, N > 320): 9 / 5 / 3 / 2 work items of 192 query rows per (batch, head); with B * 12 * items > 512 the
    workgroups walk several items (K / V tile stream and Q rows prefetched across the item boundary, ring slots and fragment
    sets carried over); against the fp32 oracle, every other forward form, and itself (bit-repeatable)."""
    KC.case_attention(DEV, torch.bfloat16, *BN)


@pytest.mark.parametrize("BN", [(24, 290), (23, 281), (22, 257), (43, 320)])
def test_attention_backward_persistent_crosses_item_boundaries(BN):
    """B * 12 > 256 (batch, head) items: workgroups of the persistent fused backward walk more than one item (the next
    item's K / V / query tiles prefetched under the current one, dK / dV stored a step late); against the fp32 autograd
    oracle and bit for bit against the one-workgroup-per-item form."""
    KC.case_attention(DEV, torch.bfloat16, *BN)


def test_split_bf16_products():
    """precision="bf16x3" kernels at model shapes: 1e-4 of the output scale against fp64."""
    e, eo = KC.case_split_precision(DEV, M=8192, N=768, K=768, B=2, Ntok=290)
    print(f"split-bf16: GEMM {e:.2e}, attention forward {eo:.2e} of the output scale")
    KC.case_split_precision(DEV, M=1024, N=2304, K=3072, B=1, Ntok=560)


def test_attention_rescale_branch():
    KC.case_attention(DEV, torch.float32, 1, 290, spike=True)
    KC.case_attention(DEV, torch.bfloat16, 1, 290, spike=True, bf16_tol=8e-2)
    # the persistent forward defers the maximum: the spike sends it down its rescale path late in the row (a raw score of ~48 against a
    # running maximum of ~4).  Its Q is rounded once more after the scale * log2(e) pre-scaling: at |score| ~ 70 (log2 units) that is
    # ~1.5e-2 on the log-sum-exp (the reference's own 16-bit autocast rounds such a score to 3e-2)
    KC.case_attention(DEV, torch.bfloat16, 2, 560, spike=True, bf16_tol=8e-2, fwd_tol=4e-2)
    # with the factor folded into the operand (MAEST_BF16_QS) q' is rounded once and the backward exponentiates the very product the
    # forward did: the ordinary forward tolerance holds on the spike as well
    KC.case_attention(DEV, torch.bfloat16, 1, 290, spike=True, bf16_tol=8e-2, qs=True)
    KC.case_attention(DEV, torch.bfloat16, 2, 560, spike=True, bf16_tol=8e-2, qs=True)


@pytest.mark.parametrize("dtype", DT)
def test_patch_embed(dtype):
    KC.case_patch_embed(DEV, dtype, 3, 626, patchout=30, mix=True)
    KC.case_patch_embed(DEV, dtype, 2, 625)
    KC.case_patch_embed(DEV, dtype, 3, 626, patchout=5, mix=True, masked=True, stride=(16, 13), seed=36)      # another patch stride


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mix", [False, True])
def test_patch_embed_spec_masking_fused(dtype, mix):
    """SpecMasking stripes as a predicate of the patch-embedding operand load (per clip, before mixup)."""
    KC.case_patch_embed(DEV, dtype, 4, 626, patchout=30, mix=mix, masked=True, seed=34)


def test_head():
    KC.case_head(DEV, 5, 281)


def test_loss():
    KC.case_loss(DEV, 64, 400)
    KC.case_loss(DEV, 7, 519)


def test_spec_mask():
    KC.case_spec_mask(DEV, 4, 626)


def test_mel():
    KC.case_mel(DEV, 2, 160000)
    KC.case_mel(DEV, 1, 480000, seed=81)
    KC.case_mel(DEV, 1, 5000, seed=82)
    KC.case_mel(DEV, 3, 40001, seed=83)      # odd clip length: clips 1, 2 start off the 8-byte grid (the element-wise fetch form inside the clip too)
    KC.case_mel(DEV, 2, 16640, seed=84)      # T = 66: one full block of 64 frames + two frames


def test_melfile(tmp_path):
    KC.case_melfile(DEV, tmp_path)
    # full-size clips (10 s = 625 frames), batch 64, ragged lengths and offsets
    rng = np.random.Generator(np.random.PCG64(5))
    counts = [int(c) for c in rng.integers(1, 1400, 64)]
    offsets = [int(rng.integers(0, max(c - 300, 1))) for c in counts]
    KC.case_melfile(DEV, tmp_path, size=625, counts=counts, offsets=offsets, seed=91)


def test_augment_mel():
    KC.case_augment_mel(DEV, 3, 320000)      # 10 s at 32 kHz -> [3, 128, 1000]
    KC.case_augment_mel(DEV, 2, 33333)


def test_swa():
    KC.case_swa(DEV)


@pytest.mark.parametrize("dtype,B,N", [(torch.bfloat16, 3, 290), (torch.bfloat16, 2, 281), (torch.bfloat16, 2, 560), (torch.float32, 2, 290), (torch.bfloat16, 1, 29)])
def test_attention_restricted_to_the_head_tokens(dtype, B, N):
    KC.case_attention_head_rows(DEV, dtype, B, N)
