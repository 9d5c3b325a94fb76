"""Streaming-rate reference for the HBM-bound passes: a device-to-device copy (hipMemcpyAsync and ATen's elementwise copy) against
bn_apply / bn_bwd_apply on the same tensor sizes.  usage: python scripts/stream_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from simhand_amd import ops


def timed(fn, reps=9):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


dt = torch.bfloat16
for m, c in ((2048 * 3136, 64), (2048 * 3136, 256), (2048 * 784, 128), (2048 * 196, 256), (2048 * 49, 512)):
    y = torch.randn(m, c, device="cuda").to(dt)
    a = torch.empty_like(y)
    da = torch.randn(m, c, device="cuda").to(dt)
    by = 2.0 * m * c
    st = ops.bn_finalize(ops.bn_partial_stats(y, m, c), m, c, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), None, None, None)
    t_copy = timed(lambda: a.copy_(y))
    t_add = timed(lambda: torch.add(y, da, out=a))
    t_app = timed(lambda: ops.bn_apply(y, st, m, c, True, None, out=a))
    t_bwd = timed(lambda: ops.bn_backward(da, None, y, st, torch.ones(c, device="cuda"), m, c, True, False, mask_from_y=True))
    print(f"[{m} x {c}] {by / 1e6:.0f} MB: copy {t_copy:.0f} us {2 * by / t_copy / 1e3:.0f} GB/s | torch add (2R+1W) {t_add:.0f} us {3 * by / t_add / 1e3:.0f} GB/s | "
          f"bn_apply (1R+1W) {t_app:.0f} us {2 * by / t_app / 1e3:.0f} GB/s | bn_backward (partial 2R, apply 2R+1W) {t_bwd:.0f} us {5 * by / t_bwd / 1e3:.0f} GB/s", flush=True)
