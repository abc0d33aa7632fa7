# The nccl (= RCCL) branch of the data-parallel path on the ONE GPU of a test box: a one-rank communicator
# (init_from_env(force=True)) and GradReducer(force_collective=True), so that every bucket all-reduce is really
# issued through librccl -- asynchronously, from the side stream the wgrad kernels run on -- and waited for in
# finish(), through three real training steps with the fused AdamW on the bucket views.  A one-rank all-reduce is
# the identity, so the run must reproduce, bit for bit, the same three steps without any collective.
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
    for it in range(3):
        red.reset()
        mod.training_step((x, None, y), it, _patchout=po).backward()
        issued += len(red._works)
        red.finish()
        if it == 0:
            g_first = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        opt.step()
        opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in net.named_parameters()}, g_first, issued, len(red.buckets)


if __name__ == "__main__":
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from maest_amd.dist import init_from_env
    assert init_from_env(backend="nccl", force=True) == (0, 0, 1)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    w1, g1, issued, nb = run(force=True)
    assert nb >= 3 and issued == 3 * nb, (issued, nb)       # every bucket of every step went through RCCL
    w0, g0, issued0, _ = run(force=False)
    assert issued0 == 0
    for n in w0:
        assert torch.equal(w0[n], w1[n]), f"weights differ with the one-rank all-reduce in the path: {n}"
    for n in g0:
        assert torch.equal(g0[n], g1[n]), f"gradients differ with the one-rank all-reduce in the path: {n}"
    dist.destroy_process_group()
    print(f"rccl one rank: {issued} bucket all-reduces over {nb} buckets x 3 steps; weights and gradients identical "
          "to the run without a collective")
