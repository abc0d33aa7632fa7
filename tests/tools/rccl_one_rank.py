# The nccl (= RCCL) branch of the data-parallel path on the ONE GPU of a test box: a one-rank communicator
# (init_from_env(force=True)) and GradReducer(force_collective=True), so that every bucket all-reduce is really
# issued through librccl -- asynchronously, from the side stream the wgrad kernels run on -- and waited for in
# finish(), through three real training steps with the fused AdamW on the bucket views.  A one-rank all-reduce is
# the identity, so the run must reproduce the same three steps without any collective -- up to the order of the
# split-K fp32 atomics of the wgrad kernels, which makes two runs of the SAME path differ in the last bits too.
import os, sys, socket
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def make(seed=5):
    from maest_amd import get_maest
    from maest_amd.module import Module
    torch.manual_seed(seed)
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").cuda().train()
    return net, Module(net=net, mixup_alpha=0.0, lr=1e-3)


def run(force):
    from maest_amd.dist import GradReducer
    net, mod = make()
    opt = mod.get_optimizer()
    red = GradReducer(net.named_parameters(), skip=("head_dist.weight", "head_dist.bias"), bucket_mb=64,
                      force_collective=force)
    assert red.collective == force
    net._grad_sink = red
    rng = np.random.Generator(np.random.PCG64(100))
    x = torch.from_numpy(rng.standard_normal((8, 1, 96, 626), dtype=np.float32)).cuda()
    y = torch.from_numpy((rng.random((8, 400)) < 0.02).astype(np.float32)).cuda()
    po = (0, torch.arange(0, 62, 2)[:32])
    issued = 0
    g_first = None
    losses = []
    w_init = {n: p.detach().clone() for n, p in net.named_parameters()}
    for it in range(3):
        red.reset()
        loss = mod.training_step((x, None, y), it, _patchout=po)
        loss.backward()
        losses.append(loss.item())
        issued += len(red._works)
        red.finish()
        if it == 0:
            g_first = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        opt.step()
        opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    w = {n: p.detach().clone() for n, p in net.named_parameters()}
    return w, g_first, issued, len(red.buckets), losses, w_init


if __name__ == "__main__":
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from maest_amd.dist import init_from_env
    assert init_from_env(backend="nccl", force=True) == (0, 0, 1)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    w1, g1, issued, nb, l1, w_init = run(force=True)
    assert nb >= 3 and issued == 3 * nb, (issued, nb)       # every bucket of every step went through RCCL
    w0, g0, issued0, _, l0, _ = run(force=False)
    assert issued0 == 0
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-4 * abs(a), (l0, l1)
    assert l1[2] < l1[0], "three AdamW steps on one batch must reduce the loss"
    worst = 0.0
    for n in g0:
        e = ((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-20)).item()
        worst = max(worst, e)
        assert e < 1e-4, f"first-step gradient differs with the one-rank all-reduce in the path: {n} ({e:.2e})"
    num = sum(((w0[n] - w1[n]).double() ** 2).sum().item() for n in w0)
    den = sum(((w0[n] - w_init[n]).double() ** 2).sum().item() for n in w0)
    wrel = (num / den) ** 0.5
    assert wrel < 2e-2, f"weights after 3 steps: ||w_rccl - w_plain|| / ||update|| = {wrel:.2e}"
    dist.destroy_process_group()
    print(f"rccl one rank: {issued} bucket all-reduces over {nb} buckets x 3 steps; losses {l1} vs {l0}; first-step gradients "
          f"equal to {worst:.1e} (atomics order), weights and gradients identical up to that: update deviation {wrel:.1e}")
