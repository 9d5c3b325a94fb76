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
