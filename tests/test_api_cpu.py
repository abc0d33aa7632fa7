"""CPU: the drop-in surface and host logic (no kernel launches): C-ABI exports, exception contract
(reference tests/test_maest.py:13-29), state_dict compatibility, loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

import maest_amd
from maest_amd import _lib, get_maest
from oracle import maest_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model():
    return get_maest(arch="discogs-maest-30s-pw-129e", pretrained=False)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "maest_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(maest_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SIGNATURES) | {"maest_version", "maest_last_error"}
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libmaest_hip.so not built here (run __graft_entry__.build())")
    import ctypes
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/maest_hip.h but not exported"
    lib.maest_version.restype = ctypes.c_int
    assert lib.maest_version() == _lib.ABI_VERSION


def test_argument_validation_returns_status_and_message():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.load()
    rc = lib.maest_gemm_nt(None, 0, None, 0, 1, None, 0, 1, 4, 4, 64, None, 0, None, None, 0, 1, None)
    assert rc == 1 and b"null operand" in lib.maest_last_error()
    rc = lib.maest_layernorm_fwd(1, 768, 1, 1, 1, 768, 0, None, None, 4, 512, 1e-6, None)
    assert rc == 1 and b"768" in lib.maest_last_error()


def test_wgrad_workspace_size_is_host_logic():
    """maest_gemm_tn_workspace_bytes (the opt-in deterministic split-K combine): one 256 KiB partial tile per workgroup of the
    split-K launch (at most one round of 256), 0 for shapes the 256-tile kernel does not take or that need no split; no device work."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    from maest_amd import ops
    tile = 256 * 256 * 4
    assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 2304, 768, 74240) == 0                       # default: fp32 atomics
    with ops.options(tn_reduce=1):
        assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 2304, 768, 74240) == 9 * 27 * tile       # qkv wgrad: 27 tiles x 9 splits
        assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 768, 768, 74240) == 28 * 9 * tile        # proj: 9 tiles x 28 splits
        assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 3072, 768, 74240) == 7 * 36 * tile
        assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 400, 768, 74240) == 0                    # ragged M: the 128-tile kernel
        assert ops.gemm_tn_workspace_bytes(torch.bfloat16, 768, 768, 74240, split_k=1) == 0         # a single split: nothing to combine
    lib = _lib.load()
    assert lib.maest_gemm_tn_workspace_bytes(1, 2304, 768, 74240, 0, None) == 1 and b"null result" in lib.maest_last_error()


def test_numpy_input(model):
    with pytest.raises(Exception):
        model(np.random.rand(128, 128))


def test_empty_input(model):
    with pytest.raises(Exception):
        model(torch.empty([]))


def test_long_2d_input(model):
    # 40 s of audio into the 30 s model: rejected (reference maest.py:664-668) before any device work
    with pytest.raises(Exception, match="reduce the input duration"):
        model(torch.rand(2, 40 * 16000).float())


def test_1d_with_melspectrogram_flag_asserts(model):
    with pytest.raises(AssertionError):
        model(torch.rand(16000), melspectrogram_input=True)


def test_cpu_tensor_fails_loudly_no_fallback(model):
    with pytest.raises(_lib.MaestHipError, match="no CPU fallback"):
        model(torch.rand(1, 96, 1875))


def test_unknown_arch_and_pretrained():
    with pytest.raises(NotImplementedError):
        get_maest("not-a-model", pretrained=False)
    with pytest.raises(RuntimeError, match="network"):
        get_maest("discogs-maest-10s-pw-129e")          # pretrained defaults to True like the reference


def test_patchout_variants_resolve_to_token_lists():
    """Every patchout variant of maest.py:678-780 is resolved on the host into the kept-token list."""
    m = get_maest("discogs-maest-10s-pw-129e", pretrained=False, s_patchout_t=20, s_patchout_f=2, u_patchout=25)
    torch.manual_seed(1)
    toff, tok = m._resolve_tokens(9, 61)
    assert tok.shape == ((9 - 2) * (61 - 20) - 25, 2) and tok.dtype == torch.int32
    flat = tok[:, 0].long() * 100 + tok[:, 1].long()
    assert bool((flat[1:] > flat[:-1]).all()), "tokens stay in f-major sequence order"
    m.eval()
    _, tok = m._resolve_tokens(9, 61)
    assert tok.shape == (9 * 61, 2)
    m2 = get_maest("discogs-maest-10s-pw-129e", pretrained=False, s_patchout_t_interleaved=2,
                   s_patchout_f_indices=(1, 8)).eval()
    _, tok = m2._resolve_tokens(9, 61)
    assert sorted(set(tok[:, 0].tolist())) == [0, 2, 3, 4, 5, 6, 7] and sorted(set(tok[:, 1].tolist())) == list(range(0, 61, 2))


def test_state_dict_matches_reference_layout():
    m = get_maest("discogs-maest-10s-pw-129e", pretrained=False)
    sd = m.state_dict()
    spec = O.state_dict_spec(625, 400)
    # the oracle's spec was probed with torchaudio stubbed out; the genuine reference additionally carries the two
    # persistent buffers torchaudio's Spectrogram / MelScale register (models/helpers/melspectrogram.py:29-42)
    mel_buffers = [("melspectrogram.spec.window", (512,)), ("melspectrogram.mel_scale.fb", (257, 96))]
    assert list(sd.keys()) == [n for n, _ in spec] + [n for n, _ in mel_buffers]
    for n, shape in spec + mel_buffers:
        assert tuple(sd[n].shape) == tuple(shape), n
    full = dict(O.make_state_dict(625))
    full.update({n: torch.zeros(s) for n, s in mel_buffers})
    m.load_state_dict(full, strict=True)         # a genuine reference state_dict (with the buffers) loads strictly
    assert float(m.melspectrogram.spec.window.abs().max()) > 0.5, "checkpoint values never replace the constant tables"
    assert sum(p.numel() for p in m.parameters()) == 85927712
    assert m.training, "get_maest returns the model in train mode like the reference (maest.py:1552)"
    m.load_state_dict(O.make_state_dict(625), strict=True)
    assert m.no_weight_decay() == {"new_pos_embed", "freq_new_pos_embed", "time_new_pos_embed", "cls_token",
                                   "dist_token"}


@pytest.mark.parametrize("arch,t,c", [("discogs-maest-5s-pw-129e", 312, 400), ("discogs-maest-20s-pw-129e", 1250, 400),
                                      ("discogs-maest-30s-pw-129e-519l", 1875, 519),
                                      ("passt_s_swa_p16_128_ap476", 998, 400)])
def test_arch_registry(arch, t, c):
    m = get_maest(arch, pretrained=False)
    assert m.img_size == (96, t) and m.num_classes == c
    assert m.time_new_pos_embed.shape == (1, 768, 1, t // 10)
    assert m.patch_embed.grid_size == (9, t // 10)
    assert len(m.labels) == c
    assert len(m.blocks) == 12


def test_init_distribution():
    torch.manual_seed(0)
    m = get_maest("discogs-maest-10s-pw-129e", pretrained=False)
    w = m.blocks[3].mlp.fc1.weight
    assert abs(float(w.std()) - 0.02) < 2e-3 and float(w.abs().max()) <= 2.0
    assert float(m.head[1].weight.std()) > 0.01          # head is trunc-normal too (maest.py:600 via apply)
    assert float(m.blocks[0].attn.qkv.bias.abs().max()) == 0.0
    assert float((m.norm.weight - 1).abs().max()) == 0.0
    assert abs(float(m.time_new_pos_embed.std()) - 0.02) < 3e-3


def test_maest_alias_package():
    from maest import get_maest as g2
    assert g2 is get_maest


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "maest_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f), encoding="utf-8").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_mixup_draw_semantics():
    from maest_amd.augment import my_mixup
    torch.manual_seed(0)
    np.random.seed(0)
    perm, lam = my_mixup(64, 0.3)
    assert sorted(perm.tolist()) == list(range(64))
    assert lam.dtype == torch.float32 and float(lam.min()) >= 0.5 and float(lam.max()) <= 1.0
    # same RNG consumption and float32 arithmetic as the reference helper (helpers/mixup.py:5-12)
    torch.manual_seed(0)
    np.random.seed(0)
    want_perm = torch.randperm(64)
    b = np.random.beta(0.3, 0.3, 64).astype(np.float32)
    want = np.concatenate([b[:, None], 1 - b[:, None]], 1).max(1)
    assert torch.equal(perm, want_perm) and np.array_equal(lam.numpy(), want)


def test_spec_masking_draw_ranges():
    from maest_amd.spec_masking import SpecMasking
    torch.manual_seed(0)
    t, f = SpecMasking().draw(8, 96, 625)
    assert t.shape == (8, 20, 2) and f.shape == (8, 8, 2)
    assert int(t[..., 1].max()) < 8 and int(f[..., 1].max()) < 5
    assert int((t[..., 0] + t[..., 1]).max()) <= 625 and int((f[..., 0] + f[..., 1]).max()) <= 96


def test_train_rng_draws_match_reference_order(model):
    """toffset is drawn before the kept-column permutation, from torch's global CPU generator."""
    m = get_maest("discogs-maest-10s-pw-129e", pretrained=False, s_patchout_t=30)
    torch.manual_seed(42)
    toff, tok = m._resolve_tokens(9, 61)
    torch.manual_seed(42)
    want_off = torch.randint(1 + 62 - 61, (1,)).item()
    want_keep = torch.randperm(61)[:31].sort().values
    assert toff == want_off and torch.equal(tok[:31, 1].long(), want_keep) and int(tok[:31, 0].max()) == 0


def test_lr_schedules_match_the_reference_fixture():
    """maest_amd/schedule.py against tests/golden/g10_lr_schedule.npz, written by oracle/gen_golden_schedule.py from the
    imported reference (helpers/ramp.py through Module.get_scheduler_lambda's argument order, models/module.py:213-226)."""
    import os
    import numpy as np
    from maest_amd.schedule import scheduler_lambda
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g10_lr_schedule.npz"))
    for tag in ("default", "b", "c"):
        w, s, l, last = g[f"exp_lin_{tag}_args"]
        kw = dict(warm_up_len=int(w), ramp_down_start=int(s), ramp_down_len=int(l), last_lr_value=float(last))
        for mode in ("exp_lin", "cos_cyc"):
            f = scheduler_lambda(schedule_mode=mode, **kw)
            got = np.array([f(int(e)) for e in g["epochs"]])
            np.testing.assert_allclose(got, g[f"{mode}_{tag}"], rtol=1e-14, atol=0, err_msg=f"{mode} {tag}")
    import pytest
    with pytest.raises(RuntimeError):
        scheduler_lambda(schedule_mode="step")


def test_configure_optimizers_returns_the_reference_layout():
    """models/module.py:245-254: {"optimizer": AdamW(lr 2e-5, wd 1e-4), "lr_scheduler": LambdaLR(exp_lin)}; the schedule
    drives the optimizer's learning rate epoch by epoch."""
    import torch
    from maest_amd.module import Module

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))
    mod = Module(net=Tiny())
    cfg = mod.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]
    assert isinstance(opt, torch.optim.AdamW) and isinstance(sched, torch.optim.lr_scheduler.LambdaLR)
    assert opt.defaults["lr"] == 2e-5 and opt.defaults["weight_decay"] == 1e-4
    f = mod.get_scheduler_lambda()
    # LambdaLR applies the epoch-0 factor on construction (exp warm-up: 0.0174 x lr), exactly as under Lightning;
    # callers that want the bare optimizer (bench.py, the parity tests) use get_optimizer()
    assert abs(opt.param_groups[0]["lr"] - 2e-5 * f(0)) < 1e-18 and f(0) < 0.02
    assert mod.get_optimizer().param_groups[0]["lr"] == 2e-5
    for epoch in range(1, 8):
        opt.step()
        sched.step()
        assert abs(opt.param_groups[0]["lr"] - 2e-5 * f(epoch)) < 1e-18
    assert isinstance(Module(net=Tiny(), adamw=False).get_optimizer(), torch.optim.Adam)


def test_explicit_patchout_draws_are_validated():
    """An offset that does not leave T' columns of the time table fails in the reference (the sliced table no longer
    broadcasts, models/maest.py:648-657); here it must raise instead of reaching the kernel, where it would read past
    the table (found through a test that passed offset 3 for a full-width input)."""
    import pytest
    import torch
    from maest_amd import get_maest
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30)
    keep = torch.arange(32)
    assert net._resolve_tokens(12, 62, pinned=(0, keep))[1].shape == (12 * 32, 2)
    assert net._resolve_tokens(12, 40, pinned=(22, None))[0] == 22
    with pytest.raises(ValueError, match="offset"):
        net._resolve_tokens(12, 62, pinned=(3, keep))
    with pytest.raises(ValueError, match="kept time columns"):
        net._resolve_tokens(12, 40, pinned=(0, torch.tensor([0, 40])))


def test_clone_carries_configuration_and_optimizer_sees_only_the_trained_net():
    """copy.deepcopy / clone_weights (what the SWA callback does, helpers/swa_callback.py:9-44) keeps what
    was changed after construction (patchout switched off, numeric mode, engine switches) and copies the parameters;
    the optimizer of a module with an SWA twin holds the trained net's parameters only (the reference creates net_swa
    after configure_optimizers)."""
    import copy
    import torch
    from maest_amd import get_maest
    from maest_amd.module import Module
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16x3")
    net.s_patchout_t = 0
    net._engine.head_tail = False
    net.eval()
    twin = copy.deepcopy(net)
    assert twin.s_patchout_t == 0 and twin.precision == "bf16x3" and twin._engine.head_tail is False and not twin.training
    for (n, a), (_, b) in zip(twin.named_parameters(), net.named_parameters()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), n
    mod = Module(net=net, do_swa=True)
    own = {id(p) for p in net.parameters()}
    held = [p for g in mod.get_optimizer().param_groups for p in g["params"]]
    assert len(held) == len(own) and all(id(p) in own for p in held)
    assert not any(id(p) in own for p in mod.net_swa.parameters())


def test_failed_register_audit_leaves_the_kernel_out_instead_of_failing_the_build(tmp_path, monkeypatch, capfd):
    """maest_amd/build.py: an owned-register source whose code object fails the audit (another hipcc's register allocation), or whose
    device assembly cannot be found, is recompiled with MAEST_OWNED_DISABLED -- the library then carries availability stubs and
    dispatches to the kernels those replaced; maest_kernel_forms() reports what is in."""
    import ctypes
    import shutil
    import subprocess
    from maest_amd import build as B, pw_audit
    if shutil.which(B._hipcc()) is None and not os.path.exists(B._hipcc()):
        pytest.skip("no hipcc here")
    build_dir = os.path.join(os.path.dirname(B.LIB), "build")
    others = [os.path.join(build_dir, s + ".o") for s in B.SOURCES if s not in B.AUDITED]
    if not all(os.path.exists(o) for o in others):
        pytest.skip("no object files of a previous build to link against")
    monkeypatch.delenv("MAEST_STRICT_AUDIT", raising=False)
    objs = []
    # (a) the audit finds an owned register outside the asm blocks; (b) no device assembly; (c) hipcc rejected the source
    monkeypatch.setattr(pw_audit, "audit", lambda *a, **k: ([(7, "owned arch VGPR outside the asm blocks", "v_mov_b32 v200, v1")], 200, {}))
    o = str(tmp_path / "attn_fwd_pw.o"); objs.append(o)
    assert "owns by hand" in B.audit_or_leave_out("attn_fwd_pw.hip", o, verbose=False)
    monkeypatch.setattr(B, "_device_asm", lambda name, bdir="build": None)
    o = str(tmp_path / "gemm_nt_ow.o"); objs.append(o)
    assert "cannot audit" in B.audit_or_leave_out("gemm_nt_ow.hip", o, verbose=False)
    o = str(tmp_path / "gemm_tn_ow.o"); objs.append(o)
    assert "rejected" in B.audit_or_leave_out("gemm_tn_ow.hip", o, compile_failed=True, verbose=False)
    err = capfd.readouterr().err
    assert err.count("WARNING") == 3 and "validated with" in err
    lib_path = str(tmp_path / "libfallback.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + others + objs)
    lib = ctypes.CDLL(lib_path)
    m = ctypes.c_int(-1)
    assert lib.maest_kernel_forms(ctypes.byref(m)) == 0 and m.value == 0
    for name in _lib.SIGNATURES:
        assert hasattr(lib, name)
    # strict mode (development): the same failure is an error
    monkeypatch.setenv("MAEST_STRICT_AUDIT", "1")
    with pytest.raises(RuntimeError, match="cannot audit"):
        B.audit_or_leave_out("gemm_nt_ow.hip", str(tmp_path / "x.o"), verbose=False)
    # and the product build has all three
    if os.path.exists(_lib.LIB_PATH):
        assert _lib.kernel_forms() == _lib.FORM_GEMM_NT_OW | _lib.FORM_GEMM_TN_OW | _lib.FORM_ATTN_FWD_PW


def test_run_ahead_bound_is_host_logic():
    """_Engine.throttle(): the recording forward of step k waits for the backward of step k - run_ahead (events stand in for the device)."""
    from maest_amd.maest import _Engine

    class Ev:
        def __init__(self, log, i):
            self.log, self.i = log, i

        def synchronize(self):
            self.log.append(self.i)

    net = get_maest("discogs-maest-10s-pw-129e", pretrained=False)
    eng = net._engine
    assert isinstance(eng, _Engine) and eng.run_ahead == 4 and eng.wgrad_wgs == 128
    waited = []
    for step in range(7):
        eng.throttle()                                   # forward of `step`
        eng._inflight.append(Ev(waited, step))           # its backward's event
    assert waited == [0, 1, 2] and len(eng._inflight) == 4
    eng.run_ahead = 0                                    # unbounded: never waits
    eng.throttle()
    assert waited == [0, 1, 2]
    eng.run_ahead = 1
    eng.throttle()
    assert waited == [0, 1, 2, 3, 4, 5, 6] and len(eng._inflight) == 0


def test_patch_strides_are_constructor_arguments_like_in_the_reference():
    """get_maest(stride_f=, stride_t=) (models/maest.py:1505-1507, 1537) warns and builds the tables of PatchEmbed.grid_size = img // stride
    (models/maest.py:234); a stride whose frequency table does not match the (F - 16) // stride + 1 patch rows fails in the forward with the
    error the reference raises from `x + self.freq_new_pos_embed` (class and message recorded from the reference: g12_patch_stride.npz) --
    before any device work, so the contract is checked here on CPU tensors."""
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_patch_stride.npz"))
    with pytest.warns(UserWarning):
        m = get_maest("discogs-maest-10s-pw-129e", pretrained=False, stride_f=16, stride_t=13)
    assert tuple(m.freq_new_pos_embed.shape) == (1, 768, 6, 1) and tuple(m.time_new_pos_embed.shape) == (1, 768, 1, 48)
    assert m.patch_embed.proj.stride == (16, 13)
    with pytest.warns(UserWarning):
        bad = get_maest("discogs-maest-10s-pw-129e", pretrained=False, stride_f=8, stride_t=10).eval()
    assert str(g["bad_stride_error_type"]) == "RuntimeError"
    with pytest.raises(RuntimeError) as e:
        bad(torch.zeros(2, 96, 626))
    assert str(e.value) == str(g["bad_stride_error_message"])
