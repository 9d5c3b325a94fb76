"""One-process-per-GPU data parallelism over RCCL (replaces the reference's
single-process nn.DataParallel, src/experiments/main.py:152-163).

* ``init_from_env`` -- reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun) and
  initialises ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" on CPU tests).
* ``allreduce_gradients`` -- bucketed SUM all-reduce of parameter gradients.  The loss is
  already normalised by the GLOBAL row count N and every rank back-propagates only
  through its own rows, so the per-rank gradients ADD up to the single-process
  gradient (no division by world size).  Buckets default to 64 MiB: xGMI rings are
  per-link bound (about 153 GB/s), a ResNet-50's 98.5 MB of fp32 gradients is two
  buckets, each about 0.7 ms on the wire.
* ``shard_pairs`` -- contiguous pair ranges per rank (both views of a pair stay together).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

HSA_IPC_ENV = "HSA_ENABLE_IPC_MODE_LEGACY"


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Returns (rank, local_rank, world_size); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault(HSA_IPC_ENV, "0")  # dmabuf IPC only on this pool
    if torch.cuda.is_available():
        # SIMHAND_SHARE_GPU=1 (tests only): several ranks on one device, which RCCL refuses -> gloo
        share = os.environ.get("SIMHAND_SHARE_GPU") == "1"
        torch.cuda.set_device(local % torch.cuda.device_count() if share else local)
        if share:
            backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class RcclComm:
    """A communicator over the C ABI's thin RCCL wrappers (``simhand_comm_*``, include/simhand_hip.h) -- the exchange steps
    of the path without torch.distributed in the data path.  Pass it as ``model.process_group`` (or as the ``group`` of
    ``ShardedNtxent`` / ``allreduce_gradients``) and the all-gathers / all-reduces of the step go RCCL -> xGMI directly on
    the current HIP stream.  One communicator per process, bound to the current device at construction."""

    _DT = {torch.float32: 0, torch.float64: 1, torch.bfloat16: 2, torch.int64: 3}
    _OP = {"sum": 0, "max": 1, "min": 2}

    def __init__(self, id_bytes: bytes, world: int, rank: int):
        import ctypes as C

        from .. import _lib

        self._lib, self._C = _lib.load(), C
        _lib.require_device()
        handle = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(id_bytes)
        _lib.check(self._lib.simhand_comm_init(buf, world, rank, C.byref(handle)), "comm_init")
        self._h = handle
        w, r = C.c_int(), C.c_int()
        _lib.check(self._lib.simhand_comm_world(self._h, C.byref(w), C.byref(r)), "comm_world")
        self.world, self.rank = w.value, r.value

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from .. import _lib

        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().simhand_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, group=None) -> "RcclComm":
        """Bootstrap over an existing torch.distributed group (any backend): rank 0's id travels by broadcast_object_list."""
        box = [cls.unique_id() if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0], dist.get_world_size(group), dist.get_rank(group))

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream().cuda_stream)  # inside `with torch.cuda.stream(s)`: s

    def side_stream(self) -> "torch.cuda.Stream":
        """The communicator's own HIP stream for collectives that overlap the compute stream (OverlappedGradReducer)."""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def all_gather_into(self, out: torch.Tensor, x: torch.Tensor) -> None:
        from .. import _lib

        assert out.is_contiguous() and x.is_contiguous() and out.numel() == x.numel() * self.world and out.dtype == x.dtype
        _lib.check(self._lib.simhand_comm_all_gather(self._h, self._C.c_void_p(x.data_ptr()), self._C.c_void_p(out.data_ptr()), x.numel(),
                                                     self._DT[x.dtype], self._stream()), "comm_all_gather")

    def all_reduce_(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        from .. import _lib

        assert t.is_contiguous()
        _lib.check(self._lib.simhand_comm_all_reduce(self._h, self._C.c_void_p(t.data_ptr()), self._C.c_void_p(t.data_ptr()), t.numel(),
                                                     self._DT[t.dtype], self._OP[op], self._stream()), "comm_all_reduce")
        return t

    def close(self) -> None:
        if self._h is not None:
            self._lib.simhand_comm_destroy(self._h)
            self._h = None


def shard_pairs(global_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """(first pair, number of pairs) of this rank; the global batch must divide evenly
    (the loss kernel's row map assumes equal shards)."""
    if global_pairs % world:
        raise ValueError(f"global batch {global_pairs} is not divisible by world size {world}")
    b = global_pairs // world
    return rank * b, b


_WIRE = {None: None, "fp32": None, "bf16": torch.bfloat16}


def _bucket_flat(tensors: List[torch.Tensor], wire) -> torch.Tensor:
    """One flat buffer holding `tensors` back to back (optionally in the reduced-precision wire format)."""
    flat = torch.cat([g.reshape(-1) for g in tensors])
    return flat if wire is None else flat.to(wire)


def _bucket_views(flat: torch.Tensor, tensors: List[torch.Tensor]) -> List[torch.Tensor]:
    """Views of the reduced bucket shaped like the tensors that went in: they BECOME the gradients (no copy back)."""
    if flat.dtype != tensors[0].dtype:
        flat = flat.to(tensors[0].dtype)
    out, off = [], 0
    for g in tensors:
        n = g.numel()
        out.append(flat[off:off + n].view_as(g))
        off += n
    return out


class OverlappedGradReducer:
    """Gradient all-reduce OVERLAPPED with the backward pass (north star: "all-reduce of gradients ... overlapped with
    backward").  The backbone is one hand-written backward (`ResNetEngine.backward`) that finishes its parameter gradients
    block by block, last stage first; each finished group is handed to ``submit``: full buckets are flattened once and go out
    as asynchronous SUM all-reduces while the earlier blocks' kernels keep the compute stream busy -- torch.distributed runs
    them on RCCL's own stream; with an ``RcclComm`` group they are issued on the communicator's side stream behind an event
    on the launch stream.  ``finish`` (end of the backbone's backward) waits and returns ``{parameter: reduced gradient}``
    whose tensors are VIEWS of the reduced buckets (nothing is copied back; the engine hands them to autograd as the
    gradients); ``reduced`` tells ``allreduce_gradients`` which parameters are done.  wire="bf16": buckets travel as bf16
    (half the ring time; the sum is then a bf16 sum).  xGMI is point to point (7 links x ~153 GB/s per GPU): a ResNet-50's
    98.5 MB of fp32 gradients are ~1-3 ms of ring time per step -- hidden behind ~80 ms of backward instead of appended."""

    def __init__(self, group=None, bucket_bytes: int = 32 << 20, wire: str = None):
        self.group, self.bucket_bytes, self.wire = group, bucket_bytes, _WIRE[wire]
        self.reduced = set()
        self._bucket: List[Tuple[object, torch.Tensor]] = []
        self._size = 0
        self._pending = []

    def active(self) -> bool:
        if isinstance(self.group, RcclComm):
            return self.group.world > 1
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _flush(self) -> None:
        if not self._bucket:
            return
        flat = _bucket_flat([g for _, g in self._bucket], self.wire)
        if isinstance(self.group, RcclComm):
            side = self.group.side_stream()
            ready = torch.cuda.Event()
            ready.record()                     # the flatten above, on the launch stream
            flat.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                self.group.all_reduce_(flat, "sum")
            work = torch.cuda.Event()
            work.record(side)
        else:
            # gloo runs its collectives on worker THREADS: with synchronised BatchNorm on, an asynchronous bucket and the BatchNorm sums
            # of the block below would be in flight at the same time.  Ranks that share one GPU over gloo (the test arrangement) showed
            # a corrupted tensor in ~2 % of such runs under GPU oversubscription (never with one collective at a time), so that
            # combination issues its buckets synchronously; RCCL (the product path) orders every collective on its stream anyway.
            from .. import ops

            serial = ops.bn_sync_active() and dist.get_backend(self.group) == "gloo" and not os.environ.get("SIMHAND_GLOO_ASYNC_BUCKETS")
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=not serial)
            if serial:
                work = None
        self._pending.append((work, flat, self._bucket))
        self._bucket, self._size = [], 0

    def submit(self, pairs) -> None:
        """pairs: iterable of (parameter, finished gradient tensor)."""
        for p, g in pairs:
            if g is None:
                continue
            self.reduced.add(id(p))
            self._bucket.append((p, g))
            self._size += g.numel() * (g.element_size() if self.wire is None else 2)
            if self._size >= self.bucket_bytes:
                self._flush()

    def finish(self) -> dict:
        self._flush()
        out = {}
        for work, flat, bucket in self._pending:
            if isinstance(work, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(work)
            elif work is not None:
                work.wait()
            for (p, _), v in zip(bucket, _bucket_views(flat, [g for _, g in bucket])):
                out[p] = v
        self._pending = []
        return out


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Parameters and buffers of rank `src` to every rank (replica consistency does not rest on identical seeding)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def enable_sync_bn(group=None) -> bool:
    """Synchronised BatchNorm over `group` (a torch.distributed group, an RcclComm, or None = the default group): every train-mode
    BatchNorm of the HIP engine and of the projection head then takes its statistics over the GLOBAL batch (ops.set_bn_sync; SURVEY
    8e "optional SyncBN": one small all-reduce of 2 C sums in the forward and one in the backward of each BatchNorm).  Off by default
    -- per-rank statistics are what the reference's DP replicas have (src/experiments/main.py:152-155).  Returns whether it is on
    (a single-rank job has nothing to synchronise)."""
    from .. import ops

    if isinstance(group, RcclComm):
        if group.world <= 1:
            ops.set_bn_sync(None)
            return False
        ops.set_bn_sync(lambda t: group.all_reduce_(t, "sum"))
        return True
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        ops.set_bn_sync(None)
        return False
    ops.set_bn_sync(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group))
    return True


def disable_sync_bn() -> None:
    from .. import ops

    ops.set_bn_sync(None)


_GRAD_PLANS: dict = {}


def reset_gradient_plans() -> None:
    """Forget the cached gradient patterns of ``allreduce_gradients`` (a model whose set of trained parameters changed)."""
    _GRAD_PLANS.clear()


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20, skip=None, wire: str = None) -> None:
    """SUM all-reduce over a FIXED parameter list: every rank buckets the same tensors in the same order, whatever
    received a gradient locally.  A trainable parameter without a gradient on this rank contributes zeros (and receives
    the other ranks' sum); parameters with requires_grad=False are skipped on every rank alike.  skip: ids of parameters an
    OverlappedGradReducer already reduced during this step's backward (the set is emptied).
    The pattern "which parameters have a gradient on ANY rank" is agreed once per parameter list (one small MAX all-reduce +
    one host read) and cached: later steps issue no host synchronisation at all.  A gradient that shows up outside the cached
    pattern raises (call reset_gradient_plans() after changing what is trained).  The reduced buckets BECOME the gradients
    (`p.grad` is re-pointed at views of them: no copy back); wire="bf16" sends them as bf16."""
    abi = isinstance(group, RcclComm)
    if abi:
        if group.world == 1:
            return
    elif not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    plist = [p for p in params if p.requires_grad and not (skip and id(p) in skip)]  # skip: already reduced during backward
    if skip:
        skip.clear()
    if not plist:
        return
    key = (tuple(id(p) for p in plist), id(group))
    keep = _GRAD_PLANS.get(key)
    if keep is None:
        has = torch.tensor([1 if p.grad is not None else 0 for p in plist], dtype=torch.int64, device=plist[0].device)
        if abi:
            group.all_reduce_(has, "max")
        else:
            dist.all_reduce(has, op=dist.ReduceOp.MAX, group=group)
        keep = _GRAD_PLANS[key] = [bool(k) for k in has.tolist()]  # the only host read, first step only
    grads: List[torch.Tensor] = []
    owners: List[torch.nn.Parameter] = []
    for p, k in zip(plist, keep):
        if not k:
            if p.grad is not None:
                raise RuntimeError("allreduce_gradients: a parameter outside the agreed gradient pattern received a gradient "
                                   "(reset_gradient_plans() after changing the trained set)")
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
        owners.append(p)
    wdt = _WIRE[wire]
    pending = []
    lo = len(grads)
    size = 0
    for i in range(len(grads) - 1, -1, -1):  # reverse registration order ~ order of production in backward
        size += grads[i].numel() * (grads[i].element_size() if wdt is None else 2)
        if size >= bucket_bytes or i == 0:
            tensors = grads[i:lo]
            flat = _bucket_flat(tensors, wdt)
            if abi:  # stream-ordered on the current stream: nothing to wait for on the host
                group.all_reduce_(flat, "sum")
                work = None
            else:
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
            pending.append((work, flat, i, lo))
            lo, size = i, 0
    for work, flat, i, hi in pending:
        if work is not None:
            work.wait()
        for p, v in zip(owners[i:hi], _bucket_views(flat, grads[i:hi])):
            p.grad = v
