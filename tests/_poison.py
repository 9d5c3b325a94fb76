"""Opt-in detector for reads of uninitialised device memory (test infrastructure; SIMHAND_POISON=<GiB>).

torch.empty hands out whatever the block held before.  A kernel that reads rows it should not (a reduction over a tile's rows past M,
a partial-sum row nobody wrote) usually sees finite leftovers and its result is still right -- until the leftover is a NaN pattern
(0 x NaN = NaN in an MFMA).  `poison(gib)` fills that many GiB of the caching allocator's large pool and a few thousand small-pool
blocks with 0xFF bytes (NaN as bf16 / fp16 / fp32, -1 as integers) and gives them back to the allocator, so every later torch.empty
returns NaNs: a dependence on uninitialised memory then fails the parity checks instead of passing by luck."""
import os


def poison(gib: float = None) -> bool:
    import torch

    if gib is None:
        gib = float(os.environ.get("SIMHAND_POISON", "0") or 0)
    if gib <= 0 or not torch.cuda.is_available():
        return False
    big = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device="cuda")
    big.fill_(0xFF)
    small = [torch.empty(1 << 19, dtype=torch.uint8, device="cuda").fill_(0xFF) for _ in range(2048)]  # the < 1 MiB pool (2-MiB blocks)
    tiny = [torch.empty(512, dtype=torch.uint8, device="cuda").fill_(0xFF) for _ in range(8192)]
    torch.cuda.synchronize()
    del big, small, tiny
    return True


_GUARDS = []  # (weak reference to the tensor handed out, its guard view)
GUARD_BYTES, GUARD_BYTE = 256, 0xA5


def guard_every() -> None:
    """SIMHAND_CANARY=1 (multi-rank workers, scripts/dist_stress.py --canary): every CONTIGUOUS device tensor that torch.empty / empty_like /
    new_empty hands out from now on is carved out of an allocation GUARD_BYTES longer, the tail filled with 0xA5; check_guards() afterwards
    names every tensor whose tail changed -- a kernel (or a stray DMA) that wrote past the end of a workspace, a partial-sum buffer or an
    output.  The library's own __device__ sinks are not covered (they are written on purpose)."""
    import weakref

    import torch

    real_empty = torch.empty

    def carve(shape, dtype, device):
        n = 1
        for s in shape:
            n *= int(s)
        es = torch.empty((), dtype=dtype).element_size()
        extra = (GUARD_BYTES + es - 1) // es
        buf = real_empty(n + extra, dtype=dtype, device=device)
        buf[n:].view(torch.uint8).fill_(GUARD_BYTE)
        t = buf[:n].view(*shape) if len(shape) else buf[:1].view(())
        _GUARDS.append((weakref.ref(t), buf[n:].view(torch.uint8), tuple(shape), dtype))
        return t

    def empty(*size, **k):
        dev = k.get("device")
        if dev is None or not str(dev).startswith("cuda") or k.get("pin_memory") or k.get("memory_format") not in (None, torch.contiguous_format):
            return real_empty(*size, **k)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return carve(shape, k.get("dtype") or torch.get_default_dtype(), dev)

    real_like = torch.empty_like

    def empty_like(t, **k):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()) or k.get("memory_format") not in (None, torch.contiguous_format, torch.preserve_format):
            return real_like(t, **k)
        return carve(tuple(t.shape), k.get("dtype") or t.dtype, k.get("device") or t.device)

    real_new = torch.Tensor.new_empty

    def new_empty(self, *size, **k):
        if not self.is_cuda and not str(k.get("device", "")).startswith("cuda"):
            return real_new(self, *size, **k)
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        return carve(shape, k.get("dtype") or self.dtype, k.get("device") or self.device)

    torch.empty, torch.empty_like, torch.Tensor.new_empty = empty, empty_like, new_empty


def check_guards() -> list:
    """-> [(shape, dtype, number of changed guard bytes)] over the guards of tensors that are still alive (a dead tensor's block may have
    been handed to somebody else)."""
    import torch

    torch.cuda.synchronize()
    bad, live = [], 0
    for ref, g, shape, dtype in _GUARDS:
        if ref() is None:
            continue
        live += 1
        n = int((g != GUARD_BYTE).sum())
        if n:
            bad.append((shape, str(dtype), n))
    return bad, live


def poison_every() -> None:
    """SIMHAND_POISON_EVERY=1 (multi-rank workers): EVERY device tensor that torch.empty / empty_like / new_empty hands out from now on is
    filled with 0xFF bytes first -- also the blocks the caching allocator re-uses, which `poison` cannot reach (they hold the previous
    tenant's finite data, so a read of rows nobody wrote passes silently until the allocator's re-use pattern shifts: with foreign
    streams in play -- record_stream defers frees by event completion -- that pattern is timing dependent).  Slow; small problems only."""
    import torch

    def wrap(fn):
        def inner(*a, **k):
            t = fn(*a, **k)
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and t.is_contiguous():
                t.view(-1).view(torch.uint8).fill_(0xFF)
            return t
        return inner

    torch.empty = wrap(torch.empty)
    torch.empty_like = wrap(torch.empty_like)
    torch.Tensor.new_empty = wrap(torch.Tensor.new_empty)
