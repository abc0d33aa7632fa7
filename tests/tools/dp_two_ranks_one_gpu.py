# The data-parallel training path (GradReducer + side stream + async all-reduce) with TWO ranks: by default both share
# the ONE GPU of a gpurun box and gloo transports the CUDA tensors; with `--backend nccl` each rank takes its own GPU
# and RCCL carries the all-reduce (needs two devices).  Checks the reduced gradients against a single-process
# computation of both half-batches, and that both ranks hold identical weights after 3 optimizer steps.
import os, sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def make(seed=5):
    from maest_amd import get_maest
    from maest_amd.module import Module
    torch.manual_seed(seed)
    net = get_maest("passt_s_swa_p16_128_ap476", pretrained=False, input_t=625, s_patchout_t=30, precision="bf16").cuda().train()
    return net, Module(net=net, mixup_alpha=0.0, lr=1e-3)


def data(rank, B=8):
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    x = torch.from_numpy(rng.standard_normal((B, 1, 96, 626), dtype=np.float32)).cuda()
    y = torch.from_numpy((rng.random((B, 400)) < 0.02).astype(np.float32)).cuda()
    return x, y


def worker(rank, world, port, out, backend):
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    from maest_amd.dist import GradReducer, broadcast_parameters, init_from_env
    r, l, w = init_from_env(backend=backend)            # the bootstrap bench.py / examples use
    assert (r, l, w) == (rank, local, world)
    net, mod = make(seed=5 + rank)                      # different init per rank: broadcast must fix it
    broadcast_parameters(net)
    opt = mod.get_optimizer()
    red = GradReducer(net.named_parameters(), skip=("head_dist.weight", "head_dist.bias"), bucket_mb=64)
    net._grad_sink = red
    x, y = data(rank)
    po = (0, torch.arange(0, 62, 2)[:32])
    g_first = None
    for it in range(3):
        red.reset()
        loss = mod.training_step((x, None, y), it, _patchout=po)
        loss.backward()
        red.finish()
        if it == 0:
            g_first = {n: p.grad.detach().clone().cpu() for n, p in net.named_parameters() if p.grad is not None}
        opt.step()
        opt.zero_grad(set_to_none=False)
    torch.cuda.synchronize()
    w = {n: p.detach().cpu() for n, p in net.named_parameters()}
    torch.save({"g": g_first, "w": w}, f"{out}/rank{rank}.pt")
    dist.destroy_process_group()


if __name__ == "__main__":
    out = "/tmp/dp_out"; os.makedirs(out, exist_ok=True)
    import socket
    with socket.socket() as sk:                      # a free rendezvous port on this box
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "gloo"
    mp.spawn(worker, args=(2, port, out, backend), nprocs=2, join=True)
    r0, r1 = torch.load(f"{out}/rank0.pt"), torch.load(f"{out}/rank1.pt")
    for n in r0["w"]:
        assert torch.equal(r0["w"][n], r1["w"][n]), f"weights diverged across ranks: {n}"
        if n in r0["g"]:
            assert torch.equal(r0["g"][n], r1["g"][n]), f"reduced gradients differ across ranks: {n}"
    # single-process reference of the first step: average of the two ranks' local gradients
    net, mod = make(seed=5)
    po = (0, torch.arange(0, 62, 2)[:32])
    acc = {}
    for rank in range(2):
        x, y = data(rank)
        net.zero_grad()
        mod.training_step((x, None, y), 0, _patchout=po).backward()
        for n, p in net.named_parameters():
            if p.grad is not None:
                acc[n] = acc.get(n, 0) + p.grad.detach().cpu() / 2
    worst = 0.0
    for n, g in acc.items():
        e = ((r0["g"][n] - g).abs().max() / g.abs().max().clamp_min(1e-12)).item()
        worst = max(worst, e)
    print(f"2 ranks ({backend}): weights identical after 3 steps; reduced gradients vs single-process mean: worst rel err {worst:.2e}")
    assert worst < 5e-3
