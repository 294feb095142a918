"""Multi-GPU layer: one process per GPU, `torch.distributed` over RCCL (backend "nccl") on xGMI.

Inference (the metric path) shards by image: rank r takes a contiguous slice of the batch, runs the
whole pass on its own GPU and exchanges nothing — the reference itself is single-device for predict
(`select_device('0,1')` still yields cuda:0, utils/torch_utils.py:202-219).  The only collectives
are control-plane: a barrier and a MAX-reduce of the elapsed time for measurement.
Training is data parallel as in the reference (engine/trainer.py:274, 286, 382-389): per-rank batch, loss * world_size
followed by DDP's mean, i.e. a SUM all-reduce of the gradients — ``allreduce_gradients`` on the trainer's flat buffer.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def single_rank_exchange() -> bool:
    """DYOLO_DDP_SINGLE_RANK=1: a ONE-rank process group still runs the whole exchange path (bucket all-reduces on device slices, the
    communication stream, the control-plane reductions) — every RCCL call of the N-rank job on one GPU (tests/test_train_gpu.py)."""
    return os.environ.get("DYOLO_DDP_SINGLE_RANK", "0") == "1"


def exchange_on() -> bool:
    """Whether gradients are exchanged at all: a process group of several ranks, or of one under ``single_rank_exchange()``."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_exchange())


def init_distributed(backend: Optional[str] = None, single_rank: bool = False) -> Tuple[int, int, int]:
    """Initialise the default process group when launched by torch.distributed.run; "nccl" is RCCL.  ``single_rank``: also for a
    world of one (bench.py --gpus 1 and the one-rank RCCL test: the barrier / MAX / SUM of the measurement go through the collective)."""
    rank, local_rank, world = dist_env()
    if (world > 1 or single_rank or single_rank_exchange()) and not dist.is_initialized():
        if backend is None:
            # DYOLO_DIST_BACKEND=gloo: rehearse an N-rank run on fewer GPUs than ranks (RCCL refuses two ranks on one device)
            backend = os.environ.get("DYOLO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                os.environ["MASTER_PORT"] = "29500"
            else:  # a lone rank picks a free port (nothing else has to find it)
                import socket

                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `n` items owned by `rank`; earlier ranks take the remainder."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_batch(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    s, e = shard_range(batch.shape[0], rank, world)
    return batch[s:e]


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (RCCL needs a device tensor, gloo a CPU one)."""
    if not dist.is_initialized():
        return value
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_flag(flag: bool, device=None, src: int = 0) -> bool:
    """Rank ``src``'s boolean on every rank — the reference's per-epoch ``dist.broadcast_object_list([self.stop], 0)`` (trainer.py:401,
    460): all ranks must leave the loop together.  One int32 device tensor over RCCL (a CPU tensor over gloo) instead of a pickled list."""
    if not dist.is_initialized():
        return bool(flag)
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.broadcast(t, src)
    return bool(int(t.item()))


def allreduce_gradients(flat_grad: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of the flat gradient buffer over all ranks (reference: loss *= world_size, trainer.py:382-383,
    then DistributedDataParallel's gradient mean — the product is the plain sum).  One bucket: for Drone-YOLO-s 43 MB fp32,
    a single ring pass over the xGMI links; BatchNorm statistics stay per rank (no SyncBN in the reference)."""
    if exchange_on():
        if flat_grad.is_cuda and dist.get_backend() != "nccl":  # gloo rehearsal: through host memory
            h = flat_grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat_grad.copy_(h)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class GradBuckets:
    """Bucketed, overlapped gradient exchange on the trainer's flat fp32 gradient buffer.

    The parameters are cut, in REVERSE registration order (the order backward produces their gradients: Detect head first,
    stem last), into ``n_buckets`` runs of about equal size.  The flat buffer is group-major (decay weights | norm weights |
    biases), each group in registration order, so a run of consecutive parameters is one contiguous slice in each group: a
    bucket = up to three slices.  A post-accumulate hook on every parameter counts its bucket down; when the last gradient of
    a bucket has landed, the bucket's slices are SUM-all-reduced asynchronously (RCCL runs on its own stream behind an event
    on the compute stream, so it waits for the kernels that wrote those gradients and overlaps the rest of backward).
    Buckets are issued strictly in order on every rank (collectives must match across ranks); ``finish()`` issues whatever
    backward did not complete and waits for all of them.  When the trainer replays the step from hipGraphs the capture is cut behind
    every bucket's flush (``cut``): K graphs, bucket k's all-reduce issued between the launches of graph k and graph k + 1, so the ring
    runs under the backward kernels of the later graphs exactly as in the eager form.  xGMI is point-to-point: a ring all-reduce moves 2(N-1)/N of the bytes
    over each link, so a few ~10 MB buckets keep the links busy without paying the per-collective latency 238 times.
    Reference: DistributedDataParallel's bucketed all-reduce behind ``trainer.py:274``; loss * world_size (trainer.py:382-383)
    followed by DDP's mean is the plain SUM used here."""

    def __init__(self, flat, n_buckets: int = 4):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.exchange = exchange_on()
        bounds, off = [], 0
        for s in flat.sizes:
            bounds.append((off, off + s))
            off += s
        group_of = {k: gi for gi, g in enumerate(flat.groups) for k in g}
        order = [k for k in flat.reg_order][::-1]
        total = sum(flat.offsets[k][1] for k in order)
        n_buckets = max(1, min(n_buckets, len(order)))
        target = total / n_buckets
        self.buckets: List[dict] = []
        cur, acc = [], 0
        for k in order:
            cur.append(k)
            acc += flat.offsets[k][1]
            if acc >= target * (len(self.buckets) + 1) and len(self.buckets) < n_buckets - 1:
                self.buckets.append({"names": cur})
                cur = []
        if cur:
            self.buckets.append({"names": cur})
        self.bucket_of = {}
        for bi, b in enumerate(self.buckets):
            rng = {}
            for k in b["names"]:
                o, c = flat.offsets[k]
                lo, hi = rng.get(group_of[k], (o, o + c))
                rng[group_of[k]] = (min(lo, o), max(hi, o + c))
                self.bucket_of[id(flat.params[k])] = bi
            b["ranges"] = [rng[g] for g in sorted(rng)]
            b["numel"] = sum(hi - lo for lo, hi in b["ranges"])
            assert b["numel"] == sum(flat.offsets[k][1] for k in b["names"]), "a bucket must be contiguous within each group"
            for k in b["names"]:
                flat.params[k].register_post_accumulate_grad_hook(lambda p: self._param_done(p))
        self.armed = False
        self.pending: List[int] = []
        self.next = 0
        self.works: list = []
        self.issued_during_backward = 0
        # sink mode (FlatState.enable_sink): gradients land in the sink, a bucket is flushed into G when its last one is in the stream
        self.sink_entries: Optional[List[torch.Tensor]] = None
        self.capturing = False  # inside a hipGraph capture: a ready bucket is flushed, and the capture is CUT behind the flush (``cut``)
        self.cut = None  # callable(bucket index), set by the trainer while it captures: ends the running capture and begins the next graph
        self._noted: dict = {}
        self._seen: set = set()
        self._done: set = set()

    # ---- sink mode ---------------------------------------------------------------------------------------------------------
    def use_sink(self) -> None:
        """Per-bucket slices of ``flat.sink_entries`` (call after ``FlatState.enable_sink``): a bucket's parameters are flushed by one
        ``dy_grad_sink_flush`` as soon as the backward function that produced its last gradient has returned."""
        rows = {k: i for i, k in enumerate(self.flat.offsets)}
        self.sink_entries = [self.flat.sink_entries[torch.tensor([rows[k] for k in b["names"]], device=self.flat.sink_entries.device)].contiguous()
                             for b in self.buckets]

    def note(self, p) -> None:
        """Listener of nn/autograd_ops.sink_armed: ``p`` = a parameter whose sink slot a backward function just took; ``None`` = that
        function has returned (its kernels are in the stream): every parameter noted since the last ``None`` is complete."""
        if p is not None:
            if id(p) not in self._seen:
                self._seen.add(id(p))
                self._noted[id(p)] = self.bucket_of.get(id(p))
            return
        done, self._noted = self._noted, {}
        for pid in done:
            self._param_done(pid)

    def _param_done(self, p) -> None:
        """The gradient of parameter ``p`` (the object, or its id) is complete and in the stream: count its bucket down — once per
        backward, whichever way the gradient came (sink slot or AccumulateGrad)."""
        pid = p if isinstance(p, int) else id(p)
        if pid in self._done or not self.pending:
            return
        self._done.add(pid)
        bi = self.bucket_of.get(pid)
        if bi is not None:
            self._ready(bi)

    def arm(self, on: bool, capturing: bool = False) -> None:
        """Call before a backward: ``on`` when that backward is followed by the optimizer step (the last micro-batch of an
        accumulation window) — earlier micro-batches only accumulate locally.  ``capturing``: the backward is being captured into a
        hipGraph — ready buckets are flushed and marked (``events``), ``exchange_after_replay`` issues the all-reduces."""
        self.armed = bool(on) and self.exchange
        self.capturing = capturing
        self.pending = [len(b["names"]) for b in self.buckets]
        self.next, self.works, self.issued_during_backward = 0, [], 0
        self._noted, self._seen, self._done = {}, set(), set()

    def _all_reduce(self, g: torch.Tensor) -> None:
        if g.is_cuda and dist.get_backend() != "nccl":
            # rehearsal of N ranks on fewer GPUs (DYOLO_DIST_BACKEND=gloo): the exchange goes through host memory
            h = g.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            g.copy_(h)
        else:
            self.works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))

    def _issue(self, bi: int) -> None:
        for lo, hi in self.buckets[bi]["ranges"]:
            self._all_reduce(self.flat.G[lo:hi])

    def _bucket_ready(self, bi: int) -> None:
        """All gradients of bucket ``bi`` are in the stream."""
        if self.sink_entries is not None:
            from . import hip_ops as H

            H.join_side_stream()  # the bucket's weight-gradient kernels run on a second stream (hip_ops.conv_wgrad_into)
            H.grad_sink_flush_(self.sink_entries[bi], self.flat.G, self.flat.S)
        if not self.capturing and self.armed:
            self._issue(bi)
            self.issued_during_backward += 1

    def _ready(self, bi: int) -> None:
        if not (self.armed or self.capturing or self.sink_entries is not None):
            return
        if not self.pending:
            return
        self.pending[bi] -= 1
        last = None
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:  # in order, identically on every rank
            self._bucket_ready(self.next)
            last = self.next
            self.next += 1
        if last is not None and self.capturing and self.cut is not None and last < len(self.buckets) - 1:
            self.cut(last)  # the graph captured so far ends with bucket `last`'s flush; what follows belongs to the next graph

    def end_backward(self) -> None:
        """After ``loss.backward()``: buckets whose parameters received no gradient this step (or whose count did not run down) are
        flushed / issued now, in order."""
        self.note(None)
        while self.next < len(self.buckets):
            self._bucket_ready(self.next)
            self.next += 1

    def exchange_upto(self, upto: int) -> None:
        """The graphed step: the graphs that end with the flushes of buckets < ``upto`` have been launched on the current stream — issue the
        all-reduces of the buckets not exchanged yet, in order.  An asynchronous RCCL all-reduce waits for what the current stream holds NOW
        (ProcessGroupNCCL records an event there) and runs on its own stream, i.e. under the graphs launched after this call."""
        if not self.armed:
            self._exchanged = upto
            return
        for bi in range(getattr(self, "_exchanged", 0), upto):
            self._issue(bi)
        self._exchanged = upto

    def begin_replay(self) -> None:
        """Before the first graph of a step is launched: the graphs flush every bucket themselves, nothing is left for end_backward."""
        self.next = len(self.buckets)
        self.works = []
        self._exchanged = 0

    def finish(self) -> None:
        if not self.exchange:
            return
        if not self.armed:  # a step without an armed backward (should not happen): one plain all-reduce
            dist.all_reduce(self.flat.G, op=dist.ReduceOp.SUM)
            return
        self.end_backward()  # (a no-op after the trainer's own end_backward / the graphed step's exchange)
        for w in self.works:
            if w is not None:
                w.wait()
        self.works, self.armed = [], False


def gather_detections(rows: torch.Tensor, counts: torch.Tensor) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """Optional host-side gather of the small (n, max_det, 6)/(n,) results to rank 0."""
    if not dist.is_initialized():
        return [(rows.cpu(), counts.cpu())]
    obj = (rows.cpu(), counts.cpu())
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
