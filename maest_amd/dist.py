"""Data-parallel gradient exchange for the training hot path: ONE process per GPU, torch.distributed
with backend "nccl" (= RCCL on ROCm) over xGMI; the only data-path collective is the gradient
all-reduce, exactly as in the reference (Lightning DDP, ex_maest.py:57 -> NCCL all-reduce of the
85.9 M parameter gradients every step).

Design for MI355X rather than a translation of DDP's reducer:
  * gradients live in ONE persistent flat fp32 buffer (343.7 MB) carved into a few LARGE buckets in
    the order the hand-written backward produces them (head -> block 11 .. block 0 -> embeddings).
    xGMI is point-to-point (7 links x ~153 GB/s per GPU) and ring collectives are per-link bound, so
    fewer / larger messages win; 288 GB of HBM makes the flat buffer free.
  * the engine's backward reports each finished parameter gradient (`on_grad`); when the last
    gradient of a bucket lands, its all-reduce is launched asynchronously and overlaps with the
    remaining backward kernels (RCCL runs on its own stream, ordered after the producer stream).
  * `finish()` waits for the buckets, averages (x 1/world, csrc/misc.hip:scale_kernel) and installs
    `param.grad` as VIEWS of the flat buffer -- no gradient copies at all.
  * parameters that receive no gradient in the current mode (`head_dist.*` when
    distilled_type == "mean") are left out of the buckets instead of emulating
    find_unused_parameters.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops


def backward_order(names: List[str]) -> List[str]:
    """Parameter names in the order `_Engine.backward` finishes their gradients."""
    def key(n):
        if n.startswith("head") or n.startswith("norm."):
            return (0, 0)
        if n.startswith("blocks."):
            return (1, -int(n.split(".")[1]))
        return (2, 0)
    return sorted(names, key=key)


class GradReducer:
    def __init__(self, named_params, bucket_mb: float = 96.0, process_group=None, skip=(),
                 force_collective: bool = False, tail_mb: float = 32.0):
        """force_collective: issue the bucket all-reduces even at world size 1 (needs an initialised process group).
        A one-rank all-reduce changes nothing numerically, but it drives the whole exchange path -- communicator, the
        asynchronous launch from the side stream, the wait in finish() -- through RCCL on a single-GPU box.
        tail_mb: cap of the LAST bucket (the one backward completes last: block 0 and the embeddings).  Its all-reduce
        cannot overlap with anything -- backward is over when it starts -- so it is kept small (one block, 29 MB, instead of
        three); the buckets before it stay large (few, long ring transfers: xGMI is per-link bound)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        params = {n: p for n, p in named_params if p.requires_grad and n not in skip}
        self.order = backward_order(list(params))
        self.params = params
        dev = next(iter(params.values())).device
        # every view starts on a 16-byte boundary (offsets padded to a multiple of 4 floats): the kernels that write gradients in
        # 16-byte pieces -- the deterministic split-K combine of the wgrad GEMM, MAEST_TN_REDUCE=1 -- then never have to fall
        # back because a 519-way head bias shifted what follows it; the pad floats stay zero and ride along in the all-reduce
        pad4 = lambda k: (k + 3) & ~3
        total = sum(pad4(p.numel()) for p in params.values())
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views: Dict[str, torch.Tensor] = {}
        self.bucket_of: Dict[str, int] = {}
        self.buckets: List[dict] = []
        cap = int(bucket_mb * 1024 * 1024 / 4)
        # the tail bucket: gradients taken from the END of the backward order while they fit under tail_mb
        tail_cap = int(min(tail_mb, bucket_mb) * 1024 * 1024 / 4)
        tail_first, acc = len(self.order), 0
        while tail_first > 1 and acc + pad4(params[self.order[tail_first - 1]].numel()) <= tail_cap:
            tail_first -= 1
            acc += pad4(params[self.order[tail_first]].numel())
        off = 0
        cur = {"start": 0, "end": 0, "names": []}
        for i, n in enumerate(self.order):
            k = params[n].numel()
            kp = pad4(k)
            if cur["names"] and ((off + kp - cur["start"]) > cap or (i == tail_first and tail_first < len(self.order))):
                cur["end"] = off
                self.buckets.append(cur)
                cur = {"start": off, "end": off, "names": []}
            self.views[n] = self.flat[off:off + k].view(params[n].shape)
            self.bucket_of[n] = len(self.buckets)
            cur["names"].append(n)
            off += kp
        cur["end"] = off
        self.buckets.append(cur)
        self._pending = [0] * len(self.buckets)
        self.reduced_this_step = 0
        self._works = []
        # timing = True: every bucket's all-reduce is bracketed by two events on the stream that launches it (the engine's dedicated
        # communication stream), launch -> complete; bucket_times() reads them back (bench.py prints them per bucket on the --gpus N line,
        # so that a scaling record can be read against DESIGN.md section 6's prediction).  Costs two event records per bucket.
        self.timing = False
        self._events = []
        self.reset()

    def reset(self):
        """Call before each backward: zero the flat buffer (wgrad kernels accumulate into it)."""
        self.flat.zero_()
        self._pending = [len(b["names"]) for b in self.buckets]
        self._works = []
        self.reduced_this_step = 0          # bucket all-reduces launched since this reset (bench.py --check-ranks, tests)

    def grad_buffer(self, name) -> Optional[torch.Tensor]:
        """Destination the engine should write `name`'s gradient into (a view of the flat buffer)."""
        return self.views.get(name)

    def note_grad(self, name) -> Optional[int]:
        """Engine callback, host-side bookkeeping only: the gradient of `name` is complete in its view.  Returns the
        index of the bucket this completes when that bucket has to be exchanged (-> reduce_bucket), else None."""
        b = self.bucket_of.get(name)
        if b is None:
            return None
        self._pending[b] -= 1
        return b if (self._pending[b] == 0 and self.collective) else None

    def reduce_bucket(self, b: int, dedicated_stream: bool = False):
        """Launch the asynchronous all-reduce of bucket `b`.  It is ordered after the stream that is current at the call:
        the caller makes that stream wait for every stream that wrote gradients of the bucket (maest.py: a dedicated
        stream that waits for the dgrad and the wgrad stream ONCE PER BUCKET -- neither of them is held up)."""
        bk = self.buckets[b]
        timed = self.timing and dedicated_stream and self.flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        w = dist.all_reduce(self.flat[bk["start"]:bk["end"]], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if timed:
            w.wait()            # a STREAM-side wait of the launching (communication) stream, which has nothing else to do; not a host wait
            e1.record()
            self._events.append((b, e0, e1))
            if len(self._events) > 64 * len(self.buckets):
                del self._events[:len(self.buckets)]
        self._works.append(w)
        self.reduced_this_step += 1

    def bucket_times(self, clear: bool = True):
        """[{bucket, mb, all_reduces, launch_to_complete_ms (mean), max_ms}] over the all-reduces timed since the last call (timing = True;
        synchronizes the device)."""
        if not self._events:
            return []
        torch.cuda.synchronize(self.flat.device)
        acc = {}
        for b, e0, e1 in self._events:
            acc.setdefault(b, []).append(e0.elapsed_time(e1))
        if clear:
            self._events = []
        return [{"bucket": b, "mb": round((self.buckets[b]["end"] - self.buckets[b]["start"]) * 4 / 2 ** 20, 1), "all_reduces": len(v),
                 "launch_to_complete_ms": round(sum(v) / len(v), 3), "max_ms": round(max(v), 3)} for b, v in sorted(acc.items())]

    def on_grad(self, name):
        """note_grad + reduce_bucket on the current stream (callers with a single stream: the host-logic tests)."""
        b = self.note_grad(name)
        if b is not None:
            self.reduce_bucket(b)

    def finish(self):
        """Wait for the exchanges, average, install param.grad views."""
        missing = [i for i, c in enumerate(self._pending) if c != 0]
        if missing:
            raise RuntimeError(f"GradReducer: buckets {missing} never completed (a gradient was not reported)")
        for w in self._works:
            w.wait()
        if self.world > 1:
            if self.flat.is_cuda:
                ops.scale_(self.flat, 1.0 / self.world)
            else:  # gloo / CPU tensors: only reached by the host-logic tests
                self.flat.div_(self.world)
        for n, p in self.params.items():
            p.grad = self.views[n]


def init_from_env(backend: Optional[str] = None, force: bool = False):
    """torch.distributed bootstrap from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world).  `force`: create the process group at
    world size 1 too (a one-rank RCCL communicator: GradReducer(force_collective=True), bench.py --force-collective)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(be, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(be, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank `src`'s weights (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            for p in module.parameters():
                dist.broadcast(p.detach(), src)      # in-place on a detached alias: bumps p's version counter
        eng = getattr(module, "_engine", None)
        if eng is not None:                          # operand copies cast from the pre-broadcast weights are stale
            eng.w.clear()
