"""CPU, world_size 2, gloo: the data-parallel gradient exchange host logic (maest_amd/dist.py) --
bucket partition in backward order, async all-reduce per completed bucket, averaging, grad views."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from maest_amd.dist import GradReducer, backward_order, broadcast_parameters, init_from_env
    r, _, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)
    names = ["cls_token", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias",
             "blocks.1.mlp.fc1.weight", "blocks.1.mlp.fc1.bias", "norm.weight", "head.1.weight", "head_dist.weight"]
    shapes = [(1, 1, 8), (8, 1, 4, 4), (24, 8), (24,), (32, 8), (32,), (8,), (5, 8), (5, 8)]
    params = [(n, torch.nn.Parameter(torch.randn(s))) for n, s in zip(names, shapes)]
    mod = torch.nn.ParameterList([p for _, p in params])
    broadcast_parameters(mod, 0)
    red = GradReducer(params, bucket_mb=0.0008, skip=("head_dist.weight",))   # ~200 floats per bucket
    order = backward_order([n for n in names if n != "head_dist.weight"])
    assert order[0] in ("norm.weight", "head.1.weight") and order[-1] in ("cls_token", "patch_embed.proj.weight")
    assert order.index("blocks.1.mlp.fc1.weight") < order.index("blocks.0.attn.qkv.weight")
    assert len(red.buckets) >= 2
    for step in range(2):
        red.reset()
        for n in red.order:
            red.grad_buffer(n).add_(float(rank + 1) * (step + 1))      # "kernel" writes the gradient
            red.on_grad(n)
        red.finish()
        want = (1 + 2) / 2 * (step + 1)
        for n, p in params:
            if n == "head_dist.weight":
                assert p.grad is None
            else:
                assert torch.allclose(p.grad, torch.full_like(p, want)), (n, p.grad.flatten()[:3], want)
                assert p.grad.data_ptr() == red.grad_buffer(n).data_ptr()    # a view, not a copy
    w0 = [p.detach().clone() for _, p in params]
    gathered = [torch.zeros_like(w0[2]) for _ in range(world)]
    dist.all_gather(gathered, w0[2])
    assert torch.equal(gathered[0], gathered[1]), "broadcast_parameters must make replicas identical"
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "worker failed"
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def test_grad_reducer_incomplete_bucket_is_an_error():
    from maest_amd.dist import GradReducer
    params = [("blocks.0.mlp.fc1.weight", torch.nn.Parameter(torch.zeros(4, 4))),
              ("norm.weight", torch.nn.Parameter(torch.zeros(4)))]
    red = GradReducer(params)
    red.reset()
    red.on_grad("norm.weight")
    with pytest.raises(RuntimeError, match="never completed"):
        red.finish()


def _eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import numpy as np
    from sklearn import metrics as skm
    from maest_amd.dist import init_from_env
    from maest_amd.module import Module
    init_from_env(backend="gloo")

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
    mod = Module(net=Tiny(), distributed_mode=True)
    # what two validation_steps on each rank would have collected (scores and labels differ per rank)
    all_y, all_s, all_l = [], [], []
    for r in range(world):
        rng = np.random.Generator(np.random.PCG64(50 + r))
        ys = [(rng.random((6, 7)) < 0.4).astype(np.float32) for _ in range(2)]
        for y in ys:
            y[0], y[1] = 1.0, 0.0
        ss = [rng.random((6, 7)).astype(np.float32) for _ in range(2)]
        ls = [float(rng.random()) for _ in range(2)]
        all_y.append(np.concatenate(ys)); all_s.append(np.concatenate(ss)); all_l.append(np.mean(ls))
        if r == rank:
            for y, s, l in zip(ys, ss, ls):
                mod.validation_outputs.append({"y": torch.from_numpy(y), "y_hat": torch.from_numpy(s), "loss": torch.tensor(l)})
    mod.on_validation_epoch_end()
    y, s = np.concatenate(all_y), np.concatenate(all_s)
    assert mod.validation_outputs == []
    assert abs(mod.logged["val_ap"] - skm.average_precision_score(y, s, average="macro")) < 1e-6
    assert abs(mod.logged["val_roc"] - skm.roc_auc_score(y, s, average="macro")) < 1e-6
    assert abs(mod.logged["val_loss"] - np.mean(all_l)) < 1e-6
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_validation_epoch_end_gathers_over_ranks_gloo():
    """models/module.py:156-202 in distributed mode: labels, scores and losses of every rank are gathered before the
    macro AP / ROC-AUC (every rank logs the metrics of the whole validation set)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, "worker failed"
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]


def _one_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from maest_amd.dist import GradReducer, init_from_env
    assert init_from_env(backend="gloo", force=True) == (0, 0, 1) and dist.is_initialized()
    params = [("blocks.0.mlp.fc1.weight", torch.nn.Parameter(torch.zeros(64, 8))),
              ("norm.weight", torch.nn.Parameter(torch.zeros(8)))]
    red = GradReducer(params, bucket_mb=0.0005, force_collective=True)
    assert red.collective and len(red.buckets) == 2
    red.reset()
    for n in red.order:
        red.grad_buffer(n).add_(3.0)
        red.on_grad(n)
    assert len(red._works) == 2          # both buckets were really exchanged
    red.finish()
    for n, p in params:
        assert torch.equal(p.grad, torch.full_like(p, 3.0))      # one rank: identity, no 1/world scaling
    plain = GradReducer(params)
    assert not plain.collective
    q.put("ok")
    dist.destroy_process_group()


def test_forced_collective_at_world_size_one_gloo():
    """GradReducer(force_collective=True) behind a one-rank process group (init_from_env(force=True)): the bucket
    all-reduces are issued and waited for although world == 1 -- the switch the GPU test uses to put RCCL on the
    record with a single device (tests/tools/rccl_one_rank.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0, "worker failed"
    assert q.get(timeout=5) == "ok"
