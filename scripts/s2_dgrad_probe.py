"""What should the four parity classes of a 3x3 / stride-2 data gradient cost?  Times the class-shaped stride-1 problems (1, 2, 2 and 4
taps over the OUTPUT grid, K = taps x cout) on the forward kernel next to the real data gradient.  usage: python scripts/s2_dgrad_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from simhand_amd import ops

dt = torch.bfloat16
N = 2048


def timed(fn, reps=7):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


lib = ops._lib_dev()
for c, h in ((128, 56), (256, 28), (512, 14)):
    ho = h // 2
    d = ops.conv_desc(N, h, h, c, c, 3, 3, 2, 1, dt)
    dy = torch.randn(N, ho, ho, c, device="cuda").to(dt)
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wt = ops.pack_crsk(w, dt)
    ops.route_reset()
    ops.conv2d_dgrad(d, dy, wt)
    routes = "+".join(k for k, v in ops.route_counts().items() if v)
    t_real = timed(lambda: ops.conv2d_dgrad(d, dy, wt))
    tot = 0.0
    parts = []
    for force in (0, 2):
        lib.simhand_test_igemm256_enable(force if force else 1)
        tot = 0.0
        parts = []
        for r, s in ((1, 1), (1, 2), (2, 1), (2, 2)):
            dd = ops.conv_desc(N, ho + r - 1, ho + s - 1, c, c, r, s, 1, 0, dt)
            x = torch.randn(N, ho + r - 1, ho + s - 1, c, device="cuda").to(dt)
            wk = ops.pack_krsc(torch.randn(c, c, r, s, device="cuda") * 0.05, dt)
            ops.route_reset()
            ops.conv2d_fwd(dd, x, wk, want_stats=False)
            rt = "+".join(k for k, v in ops.route_counts().items() if v)
            t = timed(lambda: ops.conv2d_fwd(dd, x, wk, want_stats=False))
            parts.append(f"{r}x{s}:{t:.0f}({rt})")
            tot += t
        print(f"C={c} @ {h}^2 -> dgrad {t_real:.0f} us ({routes}); class-shaped forwards [{'forced 256' if force else 'default'}]: {' '.join(parts)} = {tot:.0f} us", flush=True)
    lib.simhand_test_igemm256_enable(1)
