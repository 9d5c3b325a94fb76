"""One-process-per-GPU data parallelism over RCCL (replaces the reference's
single-process nn.DataParallel, src/experiments/main.py:152-163).

* ``init_from_env`` -- reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun) and
  initialises ``torch.distributed`` (backend "nccl" == RCCL on ROCm; "gloo" on CPU tests).  No other environment
  variable is read by this module.
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
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist

HSA_IPC_ENV = "HSA_ENABLE_IPC_MODE_LEGACY"


def init_from_env(backend: str = None, device_index: int = None) -> Tuple[int, int, int]:
    """Returns (rank, local_rank, world_size); no-op for a single process.  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT and nothing else.  device_index: the HIP device of this process when it is not LOCAL_RANK (the test arrangement
    with several ranks on one GPU passes it, together with backend="gloo": tests/_gloo_staging.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault(HSA_IPC_ENV, "0")  # dmabuf IPC only on this pool
    if torch.cuda.is_available():
        torch.cuda.set_device(local if device_index is None else device_index)
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

    def __init__(self, id_bytes: bytes, world: int, rank: int, side_id_bytes: bytes = None):
        """side_id_bytes: the id of a SECOND communicator over the same ranks, used only for collectives issued on the side
        stream (the overlapped gradient buckets).  RCCL / NCCL order the collectives of ONE communicator by their issue order on
        every rank and do not allow two of them in flight from different streams, so the bucket all-reduces that overlap the
        compute stream's own collectives (synchronised BatchNorm sums) need a communicator of their own; without one the buckets
        are issued on the compute stream (correct, not overlapped)."""
        import ctypes as C

        from .. import _lib

        self._lib, self._C = _lib.load(), C
        _lib.require_device()
        self._h = self._init(id_bytes, world, rank)
        self._h_side = self._init(side_id_bytes, world, rank) if side_id_bytes is not None else None
        w, r = C.c_int(), C.c_int()
        _lib.check(self._lib.simhand_comm_world(self._h, C.byref(w), C.byref(r)), "comm_world")
        self.world, self.rank = w.value, r.value

    def _init(self, id_bytes: bytes, world: int, rank: int):
        from .. import _lib

        handle = self._C.c_void_p()
        buf = (self._C.c_uint8 * 128).from_buffer_copy(id_bytes)
        _lib.check(self._lib.simhand_comm_init(buf, world, rank, self._C.byref(handle)), "comm_init")
        return handle

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from .. import _lib

        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().simhand_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, group=None) -> "RcclComm":
        """Bootstrap over an existing torch.distributed group (any backend): rank 0's ids (main + side-stream communicator)
        travel by broadcast_object_list."""
        box = [(cls.unique_id(), cls.unique_id()) if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(box[0][0], dist.get_world_size(group), dist.get_rank(group), side_id_bytes=box[0][1])

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream().cuda_stream)  # inside `with torch.cuda.stream(s)`: s

    def side_stream(self) -> "Optional[torch.cuda.Stream]":
        """The HIP stream of the SECOND communicator, for collectives that overlap the compute stream (OverlappedGradReducer);
        None when the communicator was built without one (collectives of one ncclComm must not be in flight from two streams)."""
        if self._h_side is None:
            return None
        if getattr(self, "_side", None) is None:
            # HIGH priority: the second communicator's kernels must become resident next to long one-block-per-CU compute kernels,
            # otherwise a peer's side-communicator kernel can wait on ours while ours waits behind the compute stream
            self._side = torch.cuda.Stream(priority=-1)
        return self._side

    def all_gather_into(self, out: torch.Tensor, x: torch.Tensor) -> None:
        from .. import _lib

        assert out.is_contiguous() and x.is_contiguous() and out.numel() == x.numel() * self.world and out.dtype == x.dtype
        _lib.check(self._lib.simhand_comm_all_gather(self._h, self._C.c_void_p(x.data_ptr()), self._C.c_void_p(out.data_ptr()), x.numel(),
                                                     self._DT[x.dtype], self._stream()), "comm_all_gather")

    def all_reduce_(self, t: torch.Tensor, op: str = "sum", side: bool = False) -> torch.Tensor:
        """side=True: through the second communicator (the caller has made side_stream() the current stream)."""
        from .. import _lib

        assert t.is_contiguous()
        h = self._h
        if side:
            if self._h_side is None:
                raise RuntimeError("RcclComm.all_reduce_(side=True): built without a side-stream communicator")
            h = self._h_side
        _lib.check(self._lib.simhand_comm_all_reduce(h, self._C.c_void_p(t.data_ptr()), self._C.c_void_p(t.data_ptr()), t.numel(),
                                                     self._DT[t.dtype], self._OP[op], self._stream()), "comm_all_reduce")
        return t

    def close(self) -> None:
        for name in ("_h_side", "_h"):
            h = getattr(self, name, None)
            if h is not None:
                self._lib.simhand_comm_destroy(h)
                setattr(self, name, None)


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


# ---- torch.distributed collectives ------------------------------------------------------------------------------------------------
# The product transport hands tensors to torch.distributed as they are: backend "nccl" (= RCCL) takes device tensors and orders every
# collective on its own stream; backend "gloo" takes the host tensors of the CPU tests.  What a test arrangement needs beyond that --
# ranks SHARING one GPU over gloo, where device tensors have to be staged through host memory -- lives in tests/_gloo_staging.py and is
# installed through set_transport(); the record-and-re-derive audit of scripts/dist_stress.py through set_collective_audit().  This
# module reads no environment variable besides the rendezvous ones of init_from_env.
class TorchTransport:
    """all_reduce_ / all_gather_into / broadcast: synchronous w.r.t. the stream semantics of the backend; bucket(): an asynchronous SUM
    all-reduce of a flat gradient bucket, returns an object with wait()."""

    def all_reduce_(self, t: torch.Tensor, rop, group) -> None:
        dist.all_reduce(t, op=rop, group=group)

    def all_gather_into(self, out: torch.Tensor, x: torch.Tensor, group) -> None:
        dist.all_gather_into_tensor(out, x, group=group)

    def broadcast(self, t: torch.Tensor, src: int, group) -> None:
        dist.broadcast(t, src=src, group=group)

    def bucket(self, flat: torch.Tensor, group):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)


_TRANSPORT = TorchTransport()


def set_transport(transport: "Optional[TorchTransport]" = None) -> None:
    """Replace how this module hands tensors to torch.distributed (None = the product transport).  Tests only."""
    global _TRANSPORT
    _TRANSPORT = transport if transport is not None else TorchTransport()


_REDUCE_OPS = {"sum": "SUM", "max": "MAX", "min": "MIN"}


def all_reduce_(t: torch.Tensor, op: str = "sum", group=None) -> torch.Tensor:
    """In-place all-reduce of a contiguous tensor over `group` (RcclComm, torch.distributed group or None = default group),
    stream-ordered on the current stream."""
    if isinstance(group, RcclComm):
        return group.all_reduce_(t, op)
    _TRANSPORT.all_reduce_(t, getattr(dist.ReduceOp, _REDUCE_OPS[op]), group)
    return t


def all_gather_into(out: torch.Tensor, x: torch.Tensor, group=None) -> torch.Tensor:
    """out [world * x.numel()] <- every rank's contiguous x in rank order."""
    if isinstance(group, RcclComm):
        group.all_gather_into(out, x)
        return out
    _TRANSPORT.all_gather_into(out, x, group)
    return out


# Observer hook: an object with note(tag, input clone, result tensor, owners) sees every collective this module issues (the audit of
# tests/_gloo_staging.py keeps the inputs and re-derives each result on the host afterwards).  None = off (the product never sets it).
_AUDIT = None


def set_collective_audit(audit=None) -> None:
    global _AUDIT
    _AUDIT = audit


class OverlappedGradReducer:
    """Gradient all-reduce OVERLAPPED with the backward pass (north star: "all-reduce of gradients ... overlapped with
    backward").  The backbone is one hand-written backward (`ResNetEngine.backward`) that finishes its parameter gradients
    block by block, last stage first; each finished group is handed to ``submit``: full buckets are flattened once and go out
    as asynchronous SUM all-reduces while the earlier blocks' kernels keep the compute stream busy -- torch.distributed (nccl)
    runs them on RCCL's own stream; with an ``RcclComm`` group they are issued on the side stream of the communicator's SECOND
    ncclComm behind an event on the launch stream (one ncclComm must not have collectives in flight from two streams: the compute
    stream's own collectives -- synchronised BatchNorm, the loss exchanges -- keep the first); over gloo (ranks sharing a GPU:
    tests) through the staging transport of tests/_gloo_staging.py.  ``finish`` (end of the backbone's backward) waits and returns ``{parameter: reduced
    gradient}`` whose tensors are VIEWS of the reduced buckets (nothing is copied back; the engine hands them to autograd as the
    gradients); ``reduced`` tells ``allreduce_gradients`` which parameters are done.  wire="bf16": buckets travel as bf16
    (half the ring time; the sum is then a bf16 sum).  xGMI is point to point (7 links x ~153 GB/s per GPU): a ResNet-50's
    98.5 MB of fp32 gradients are ~1-3 ms of ring time per step -- hidden behind ~80 ms of backward instead of appended."""

    def __init__(self, group=None, bucket_bytes: int = 32 << 20, wire: str = None):
        self.group, self.bucket_bytes, self.wire = group, bucket_bytes, _WIRE[wire]
        self.reduced = set()
        self._bucket: List[Tuple[object, torch.Tensor]] = []
        self._size = 0
        self._pending = []
        self.side_buckets = 0  # buckets that went out on the side stream of an RcclComm's second communicator (overlapped)

    def active(self) -> bool:
        if isinstance(self.group, RcclComm):
            return self.group.world > 1
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _flush(self) -> None:
        if not self._bucket:
            return
        flat = _bucket_flat([g for _, g in self._bucket], self.wire)
        if _AUDIT is not None:
            _AUDIT.note(f"bucket{len(self._pending)}", flat.clone(), flat, [(id(p), g.numel()) for p, g in self._bucket])
        if isinstance(self.group, RcclComm):
            from .. import ops

            # Under synchronised BatchNorm the compute stream issues collectives of the FIRST communicator all through the backward;
            # two communicators with kernels in flight at once are only deadlock-free if both can be resident on every rank, which
            # nothing guarantees next to one-block-per-CU compute kernels, and the combination has no multi-GPU run behind it
            # (open issue): buckets then go out in order on the launch stream (correct, not overlapped).
            side = None if ops.bn_sync_active() else self.group.side_stream()
            if side is None:                       # no second communicator: in order on the launch stream
                self.group.all_reduce_(flat, "sum")
                work = None
            else:
                ready = torch.cuda.Event()
                ready.record()                     # the flatten above, on the launch stream
                flat.record_stream(side)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    self.group.all_reduce_(flat, "sum", side=True)
                self.side_buckets += 1
                work = torch.cuda.Event()
                work.record(side)
        else:
            work = _TRANSPORT.bucket(flat, self.group)
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
            _TRANSPORT.broadcast(t.data, src, group)


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

    def _sync(t: torch.Tensor) -> None:
        if _AUDIT is not None:
            pre = t.clone()
            all_reduce_(t, "sum", group)
            _AUDIT.note("bn_sync", pre, t.clone())
        else:
            all_reduce_(t, "sum", group)

    ops.set_bn_sync(_sync)
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
        all_reduce_(has, "max", group)
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
            if _AUDIT is not None:
                _AUDIT.note(f"tail{len(pending)}", flat.clone(), flat, [(id(q), t.numel()) for q, t in zip(owners[i:lo], tensors)])
            if abi:  # stream-ordered on the current stream: nothing to wait for on the host
                group.all_reduce_(flat, "sum")
                work = None
            else:
                work = _TRANSPORT.bucket(flat, group)
            pending.append((work, flat, i, lo))
            lo, size = i, 0
    for work, flat, i, hi in pending:
        if work is not None:
            work.wait()
        for p, v in zip(owners[i:hi], _bucket_views(flat, grads[i:hi])):
            p.grad = v
