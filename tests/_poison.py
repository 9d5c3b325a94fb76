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
